#!/usr/bin/env python3
"""Time the BEV-histogram metrics (rangeldm_amd/csrc/metrics.hip) at the reference's evaluation size (1000 generated vs
1000 real sweeps, 100 x 100 bins) and price the spectral-norm kernel against the fp32 VALU peak.

    python tools/bench_metrics.py [--n 1000] [--points 60000]

Algorithmic FLOPs of one pair: (1 Gram + 12 squarings) x 2 x 100^3 = 26 MFLOP (fp32 FMA); pairs = n(n-1) + n^2.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_VALU_TFLOPS = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--points", type=int, default=60000)
    a = ap.parse_args()
    from rangeldm_amd import metrics as M
    rng = np.random.default_rng(1)
    g = torch.Generator(device="cuda").manual_seed(1)
    clouds = []
    for i in range(a.n * 2):
        r = torch.randn(a.points, generator=g, device="cuda").abs() * (18.0 + 8.0 * (i >= a.n)) + 1.0
        az = torch.rand(a.points, generator=g, device="cuda") * 6.2831853 - 3.14159265
        clouds.append(torch.stack([r * az.cos(), r * az.sin(), torch.zeros_like(r), torch.zeros_like(r)], 1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hx = M.point_cloud_to_histogram(160, 100, clouds[:a.n], 3.0, 70.0)
    hy = M.point_cloud_to_histogram(160, 100, clouds[a.n:], 3.0, 70.0)
    torch.cuda.synchronize()
    t_hist = time.perf_counter() - t0
    t0 = time.perf_counter()
    jsd = M.jsd_2d(hx, hy)
    t_jsd = time.perf_counter() - t0
    M.compute_mmd(hx[:8], hy[:8])
    t0 = time.perf_counter()
    terms = M.compute_mmd(hx, hy, return_terms=True)
    t_mmd = time.perf_counter() - t0
    pairs = a.n * (a.n - 1) + a.n * a.n          # two strict upper triangles + the cross table
    flops = pairs * 13 * 2 * 100 ** 3
    print(json.dumps({"n": a.n, "points_per_cloud": a.points, "histogram_s": round(t_hist, 4),
                      "histogram_Mpoints_per_s": round(2 * a.n * a.points / t_hist / 1e6, 1), "jsd_s": round(t_jsd, 5),
                      "mmd_s": round(t_mmd, 4), "pairs": pairs, "mmd_fp32_tflops": round(flops / t_mmd / 1e12, 1),
                      "frac_of_fp32_valu_peak": round(flops / t_mmd / 1e12 / PEAK_FP32_VALU_TFLOPS, 4),
                      "jsd": jsd, "mmd_terms": terms}))


if __name__ == "__main__":
    main()
