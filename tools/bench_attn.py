#!/usr/bin/env python3
"""Times the fused GroupNorm -> q/k/v -> softmax.V launch alone at the UNet's geometries (batch 16):
    python tools/bench_attn.py            # env RLDM_ATTN_OLD=1 / RLDM_ATTN_PF=0 / RLDM_ATTN_HG=n select variants
A launch in isolation is back-to-back with itself (no cold inputs), so these are lower bounds of the in-graph numbers."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from rangeldm_amd import _lib  # noqa: E402

_lib.require_gpu()
L = _lib.lib()
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("RLDM_ATTN"))
for (B, Lt, Cc) in ((16, 1024, 128), (16, 256, 256), (16, 64, 256), (4, 512, 128), (4, 128, 256), (4, 32, 256)):
    us = C.c_float()
    _lib.check(L.rldm_bench_attention_qkv(B, Lt, Cc, 20, 200, C.byref(us), _lib.stream_ptr(torch.device("cuda"))), "bench")
    fl = 4.0 * B * (Cc // 8) * Lt * Lt * 8 + 2.0 * B * Lt * 3 * Cc * Cc
    print(f"[{tag or 'default'}] B={B} L={Lt} C={Cc}: {us.value:7.2f} us  {fl / us.value / 1e6:7.1f} TFLOP/s")
