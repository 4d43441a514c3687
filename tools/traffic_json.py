#!/usr/bin/env python3
"""profiles/round1_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the bench command.

HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and on gfx950 FETCH_SIZE
reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported.  Kernel template
instantiations are mapped to the op names bench.py reports.

usage: tools/traffic_json.py <fetch-pass-dir> <write-pass-dir> > profiles/round1_traffic.json
"""
import collections
import csv
import glob
import json
import re
import sys


def op_name(k):
    k = re.sub(r"^void\s+", "", k).replace("rldm::", "")
    m = re.match(r"conv_igemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", k)
    if m:
        v = list(map(int, m.groups()))
        return f"conv_igemm_kernel<{v[1]},{v[2]},CK{v[6]},taps{v[7]}>"
    m = re.match(r"conv_small_kernel<(\d+), (\d+), (\d+), (\d+)>", k)
    if m:
        nwn, _, taps, mi = map(int, m.groups())
        return f"conv_small_kernel<{32 * mi},{32 * nwn},taps{taps}>"
    m = re.match(r"conv_stream_kernel<(\d+), (\d+)(?:, (\d+))?(?:, (\d+))?(?:, (\d+))?(?:, (true|false))?(?:, (true|false))?>", k)
    if m:       # (round 4: third argument = waves, fourth = 32-pixel fragments per wave; the 4- and 8-wave instances of one tile share bench.py's name;
                #  sixth = the sub-pixel form of nearest x2 + 3x3)
        wm, wn, mi = int(m.group(1)), int(m.group(2)), int(m.group(4) or 4)
        return (f"conv_stream_kernel<{32 * mi * wm},{32 * wn},CK64,taps9{',s2' if (m.group(5) or '1') == '2' else ''}"
                f"{',sub' if m.group(6) == 'true' else ''}>")
    m = re.match(r"conv_regw_kernel<(\d+), (\d+)>", k)
    if m:
        return f"conv_regw_kernel<128,{32 * int(m.group(2))},taps9>"
    m = re.match(r"trunk_kernel<(\d+)>", k)
    if m:       # persistent launches: one entry per kernel variant (bench.py scales it by a launch's share of the variant's phases)
        return "trunk_kernel<" + ("conv_small image tiles", "conv_small 64x64 clusters", "conv_stream 256x128", "conv_stream 128x64",
                                 "conv_stream 128x128 x2/CU", "conv_stream 64x128")[int(m.group(1))] + ">"
    if re.match(r"attention_qkv2_d8_kernel<1>", k):      # bench.py's name of the 1024-token launch with its fused output projection
        return "attention_qkv_d8_kernel + to_out"
    if re.match(r"attention_qkv2_d8_kernel<0>", k):
        return "attention_qkv_d8_kernel"
    return re.sub(r"\(.*$", "", k)


def collect(d, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = op_name(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {"_note": "HBM-side bytes per launch from two rocprofv3 --pmc passes over `python bench.py --steps 1 --warmup 1 "
                    "--no-cpu-baseline` (FETCH_SIZE, WRITE_SIZE; counters are in KiB). bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: "
                    "on gfx950 FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM); WRITE_SIZE "
                    "uncorrected. Raw per-kernel values under _raw_kib."}
    raw = {}
    for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
        if not (k.startswith("conv_") or k.startswith("attention") or k.startswith("gn_") or k.startswith("trunk_")):
            continue
        f, nf = fetch[k]
        w = write.get(k, (0.0, 0))[0]
        out[k] = int((2 * f + w) * 1024)
        raw[k] = {"FETCH_SIZE": round(f, 1), "WRITE_SIZE": round(w, 1), "dispatches": nf}
    out["_raw_kib"] = raw
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
