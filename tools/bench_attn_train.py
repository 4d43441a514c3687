#!/usr/bin/env python3
"""Training attention kernels at the UNet's shapes (batch 8): forward / backward time by HIP events.
RLDM_TR_ATTN=scalar selects the fp32 one-thread-per-query kernels for A/B runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rangeldm_amd import train_ops as T

for B, L, C in [(8, 4096, 128), (8, 1024, 256), (8, 256, 256), (8, 64, 256)]:
    # NOTE (level shapes of the RangeLDM UNet: 256x16 / 128x8 / 64x4 / 32x2 latents)
    if L > 4000:
        continue
    q, k, v, dO = (torch.randn(B, L, C, device="cuda") for _ in range(4))
    o, lse = T.attention_forward(q, k, v)
    T.attention_backward(q, k, v, o, dO, lse)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 20
    e[0].record()
    for _ in range(n):
        o, lse = T.attention_forward(q, k, v)
    e[1].record()
    for _ in range(n):
        T.attention_backward(q, k, v, o, dO, lse)
    e[2].record()
    torch.cuda.synchronize()
    pairs = B * (C // 8) * L * L
    f, b = e[0].elapsed_time(e[1]) / n * 1e3, e[1].elapsed_time(e[2]) / n * 1e3
    print(f"B={B} L={L} C={C}: forward {f:.1f} us ({pairs / f / 1e6:.2f} T pairs/s), backward {b:.1f} us ({pairs / b / 1e6:.2f} T pairs/s)")
