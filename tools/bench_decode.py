#!/usr/bin/env python3
"""One VAE decode of a latent batch, timed with HIP events (synthetic weights): tools/bench_decode.py [--B 16] [--preset RangeLDM]."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--preset", default="RangeLDM")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    p, unet, vae, _, _ = bench.build_models(a.preset, 0)
    del unet
    shape = (p["vae"].z_channels,) + tuple(s // p["vae"].downscale for s in p["vae"].sample_size)
    z = torch.randn(a.B, *shape, device="cuda")
    for _ in range(3):
        vae.decode(z)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        vae.decode(z)
    e1.record()
    torch.cuda.synchronize()
    print(f"decode of {a.B} x {shape}: {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us")


if __name__ == "__main__":
    main()
