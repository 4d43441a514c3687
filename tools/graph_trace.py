#!/usr/bin/env python3
"""Timeline of the UNet launches as they run back to back inside the sampler's captured step graph.

A host-side profiler (rocprofv3 --kernel-trace) spaces the kernels out; here a one-thread kernel writes the 100 MHz
real-time counter between consecutive launches of the graph itself (rldm_debug_set_flags(8192) before the sampler is
built), so the deltas are the durations the launches have in the production regime (+ the constant cost of the stamp
launch, printed as the median gap of the cheapest op).

usage: python tools/graph_trace.py [--flags N] [--batch 16] [--top 40]
"""
import argparse
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rangeldm_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=0, help="extra routing flags (see rldm_debug_set_flags)")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10, help="sampler steps (the trace keeps the last one)")
    ap.add_argument("--top", type=int, default=200)
    ap.add_argument("--preset", default="RangeLDM", help="RangeLDM | nuscenes | RangeDM (config.PRESETS)")
    a = ap.parse_args()
    _lib.require_gpu()
    _lib.lib().rldm_debug_set_flags(8192 | a.flags)
    from rangeldm_amd.pipelines import LDMPipelineRange, DDIMPipelineRange
    from rangeldm_amd.schedulers import DDIMSchedulerHIP
    from rangeldm_amd.synth import latent_noise
    dev = torch.device("cuda", 0)
    p, unet, vae, _, _ = bench.build_models(a.preset, 20240310)
    if vae is not None:
        pipe = LDMPipelineRange(vae=vae, unet=unet, scheduler=DDIMSchedulerHIP(), pos_encoding=p["pos_encoding"])
    else:       # (pixel space: RangeDM)
        pipe = DDIMPipelineRange(unet=unet, scheduler=DDIMSchedulerHIP(), pos_encoding=p["pos_encoding"])
    shape = (p["unet"].out_channels, *p["unet"].sample_size)
    x = torch.from_numpy(np.stack([latent_noise(1, j, shape) for j in range(a.batch)])).to(dev)
    for _ in range(2):
        pipe(batch_size=a.batch, num_inference_steps=a.steps, latents=x, output_type="torch")
    torch.cuda.synchronize()
    stamps = (C.c_ulonglong * 4096)()
    names = C.create_string_buffer(1 << 16)
    n = _lib.lib().rldm_debug_graph_trace(stamps, 4096, names, len(names))
    assert n > 0, "no trace (library built without the stamp hook?)"
    nm = names.value.decode().split("\n")[:n]
    t = np.array([stamps[i] for i in range(n + 1)], dtype=np.int64)
    d = (t[1:] - t[:-1]) * 0.01          # us (100 MHz)
    print(f"{n} launches, {d.sum():.1f} us per UNet forward inside the graph (stamp launches included)")
    agg = collections.OrderedDict()
    for name, v in zip(nm, d):
        k = agg.setdefault(name, [0, 0.0])
        k[0] += 1
        k[1] += v
    print(f"{'kernel':48s} {'n':>4s} {'total_us':>9s} {'avg_us':>8s}")
    for name, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:48s} {c:4d} {v:9.1f} {v / c:8.2f}")
    if a.top:
        print("\nper launch, in order:")
        for i, (name, v) in enumerate(zip(nm, d)):
            if i < a.top:
                print(f"{i:4d} {name:48s} {v:8.2f}")


if __name__ == "__main__":
    main()
