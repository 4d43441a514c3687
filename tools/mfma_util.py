#!/usr/bin/env python3
"""MFMA utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass summarised by
tools/pmc_summary.py: busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs, so
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024).
usage: tools/mfma_util.py gpurun_out/<summary>.txt > profiles/<name>_mfma_util.txt"""
import sys

rows, name, vals = [], None, {}
for line in open(sys.argv[1]):
    if not line.startswith(" "):
        if name and len(vals) == 2:
            rows.append((name, vals))
        name, vals = line.strip(), {}
    else:
        p = line.split()
        vals[p[0]] = (float(p[1]), int(p[2].strip("()n=")))
if name and len(vals) == 2:
    rows.append((name, vals))
out = []
for n, v in rows:
    busy, cnt = v["SQ_VALU_MFMA_BUSY_CYCLES"]
    act, _ = v["GRBM_GUI_ACTIVE"]
    if busy <= 0:
        continue
    out.append((busy * cnt, n, cnt, act / 8, busy, busy / (act / 8 * 1024)))
print(f"{'kernel':72s} {'launches':>8s} {'cycles/launch':>14s} {'MFMA busy (SIMD-cycles)':>24s} {'MFMA util':>10s}")
for _, n, cnt, cyc, busy, u in sorted(out, reverse=True):
    print(f"{n[:72]:72s} {cnt:8d} {cyc:14.0f} {busy:24.0f} {u:10.1%}")
