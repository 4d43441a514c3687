#!/usr/bin/env python3
"""A/B of the all-taps weight-gradient launch at the 256x16 level (batch 8, 128 -> 128, 3x3): the plain instance on a materialised
activated input against the fused instance (GroupNorm + SiLU rebuilt while staging) and the fused instance without a GroupNorm.
Measured: 33 - 35 / 42.8 / 33.9 us (launch + reduction, HIP events, operands cache-hot).  usage: python tools/wgrad_ab.py"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rangeldm_amd import train_ops as T
B, C, W, H, N = 8, 128, 256, 16, 128
g = torch.Generator().manual_seed(0)
x = torch.randn(B, W, H, C, generator=g).cuda()
dy = torch.randn(B, W, H, N, generator=g).cuda()
gamma, beta = torch.ones(C).cuda(), torch.zeros(C).cuda()
cs = T.chan_stats(x)
gn = T.GN(gamma, beta, True, 32, 1e-5)
h, st = T.gn_forward(x, gamma, beta, 32, 1e-5, True)
dw = torch.zeros(N, C, 3, 3).cuda()
srcs = [T.Src(x, cs)]
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("wgrad_bias (materialised h)  %.1f us" % timeit(lambda: T.wgrad_bias(dy, h, dw, 9)))
print("wgrad_fused (gn rebuilt)     %.1f us" % timeit(lambda: T.wgrad_fused(dy, srcs, dw, 9, gn=gn)))
print("wgrad_fused (no gn, plain x) %.1f us" % timeit(lambda: T.wgrad_fused(dy, [T.Src(h)], dw, 9)))
print("wgrad_bias again             %.1f us" % timeit(lambda: T.wgrad_bias(dy, h, dw, 9)))
