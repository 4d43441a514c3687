#!/usr/bin/env python3
"""Is the training step host-bound?  Host time to ENQUEUE a forward + backward vs. GPU time to execute it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rangeldm_amd.config import UNetConfig
from rangeldm_amd.params import unet_param_shapes
from rangeldm_amd.synth import synth_state_dict
from rangeldm_amd import training as TR, train_ops as T
cfg = UNetConfig()
tr = TR.UNetTrainer(cfg, synth_state_dict(unet_param_shapes(cfg)), use_ema=True)
x = torch.randn(8, 4, 256, 16).cuda(); tgt = torch.randn(8, 4, 256, 16).cuda(); t = torch.randint(0, 1000, (8,)).cuda()
for i in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(x, t, tgt, pos_encoding=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"iter {i}: enqueue {1e3 * (t1 - t0):.2f} ms, total {1e3 * (t2 - t0):.2f} ms")
