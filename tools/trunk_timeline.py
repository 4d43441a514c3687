#!/usr/bin/env python3
"""In-kernel timeline of one persistent trunk launch (ABLATE build): s_memtime stamps of workgroup 0, wave 0, per phase.
usage: RLDM_LIB=$PWD/rangeldm_amd/librangeldm_hip_ablate.so RLDM_TS_TRUNK=<phases of the launch to stamp> python tools/trunk_timeline.py [--B 16]
Stamps per phase: 0 entry, 1 arguments, 2 small requests issued, 3 (same), 4 cluster published, 5.. as conv_small (ring, tile loads
issued, zero rows, tile stored, barrier, K loop, barrier, partials, output + statistics, end, after arrive)."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rangeldm_amd import _lib
from rangeldm_amd.config import UNetConfig
from rangeldm_amd.synth import synth_state_dict
from rangeldm_amd.params import unet_param_shapes
from rangeldm_amd.unet import UNet2DModelHIP

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=16)
a = ap.parse_args()
_lib.require_gpu()
_lib.lib().rldm_debug_timestamps(None)
cfg = UNetConfig()
m = UNet2DModelHIP(cfg)
m.load_state_dict(synth_state_dict(unet_param_shapes(cfg), prefix="tt."))
x = torch.randn(a.B, cfg.in_channels, *cfg.sample_size, device="cuda")
for _ in range(3):
    m(x, 300)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
_lib.lib().rldm_debug_timestamps(buf)
t0 = None
for ph in range(16):
    v = [buf[ph * 16 + i] for i in range(16)]
    v = [t for t in v if t]
    if not v:
        continue
    if t0 is None:
        t0 = v[0]
    print(f"phase {ph:2d}: start +{v[0] - t0:7d} | " + " ".join(f"{b - a_:5d}" for a_, b in zip(v, v[1:])) + f" | total {v[-1] - v[0]}")
