#!/usr/bin/env python3
"""Time the range-image <-> point-cloud kernels (rangeldm_amd/csrc/lidar.hip) at the BASELINE config-2 output size and
price them against HBM: one JSON line per entry point with algorithmic bytes, achieved GB/s and fraction of 8 TB/s.

    python tools/bench_lidar.py [--batch 16] [--iters 50]

Algorithmic bytes (each operand touched once):
  to_pc_torch   read 2 x 4 B, write 16 B per pixel                         = 24 B / pixel
  to_voxel      read 8 B per pixel + write the (2, 1024, 1024) fp32 volume = 8 B / pixel + 8 MiB / image
  filter_points read + write 16 B per point (upper bound: every point kept) = 32 B / point
  render_u8     read 4 B, write 1 B per pixel                              = 5 B / pixel
  project       read 16 B per return + write (2 x 4 + 2) B per pixel
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0


def timed(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    from rangeldm_amd import range_image as RI
    rng = np.random.default_rng(3)
    B, W, H = a.batch, 1024, 64
    metres = rng.uniform(1.0, 110.0, (B, W, H)).astype(np.float32)
    img = torch.from_numpy(np.stack([(metres - 20) / 40, rng.uniform(0, 1, (B, W, H)).astype(np.float32)], 1)).cuda()
    t = RI.point_cloud_to_range_image_KITTI()
    px = B * W * H
    pc = t.to_pc_torch(img)
    vox = t.to_voxel(img)
    sweep = pc[0].contiguous()
    rows = []

    def report(name, sec, nbytes, units, unit_name):
        gbs = nbytes / sec / 1e9
        rows.append({"op": name, "us": round(sec * 1e6, 2), "alg_bytes": int(nbytes), "GB/s": round(gbs, 1),
                     "frac_of_hbm_peak": round(gbs / PEAK_HBM_GBS, 4), f"{unit_name}/s": round(units / sec, 1)})

    report("to_pc_torch", timed(lambda: t.to_pc_torch(img), a.iters), 24 * px, B, "images")
    report("to_voxel", timed(lambda: t.to_voxel(img), a.iters), 8 * px + B * vox[0].numel() * 4, B, "images")
    report("filter_points", timed(lambda: t.filter_points(pc, 90.0), a.iters), 32 * px, B, "images")
    report("render_u8(range)", timed(lambda: RI.render_u8(img), a.iters), 5 * px, B, "images")
    report("render_u8(bev)", timed(lambda: RI.render_u8(vox), a.iters), 5 * B * 1024 * 1024, B, "images")
    report("project", timed(lambda: t.project(sweep), a.iters), 16 * sweep.shape[0] + 10 * W * H, 1, "sweeps")
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
