#!/usr/bin/env python3
"""Time of one UNet forward at batch B (HIP events over N replays of the eager plan), values ignored: for ablation builds whose
results are garbage (RLDM_LIB=... python tools/fwd_time.py [--B 16] [--n 50])."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rangeldm_amd import _lib
from rangeldm_amd.config import UNetConfig
from rangeldm_amd.synth import synth_state_dict
from rangeldm_amd.params import unet_param_shapes
from rangeldm_amd.unet import UNet2DModelHIP

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=16)
ap.add_argument("--n", type=int, default=50)
ap.add_argument("--size", type=str, default="")
a = ap.parse_args()
_lib.require_gpu()
cfg = UNetConfig(sample_size=tuple(int(v) for v in a.size.split("x"))) if a.size else UNetConfig()
m = UNet2DModelHIP(cfg)
m.load_state_dict(synth_state_dict(unet_param_shapes(cfg), prefix="ft."))
x = torch.randn(a.B, cfg.in_channels, *cfg.sample_size, device="cuda")
for _ in range(5):
    m(x, 300)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.n):
        m(x, 300)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / a.n)
print(f"B={a.B}: {best * 1e3:.1f} us per forward ({m.num_launches(a.B)} launches)")
