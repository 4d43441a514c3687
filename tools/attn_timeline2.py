#!/usr/bin/env python3
"""Per-wave phase stamps of the fused attention launch ALONE (ABLATE build: make -C rangeldm_amd/csrc ABLATE=1):
    RLDM_LIB=rangeldm_amd/librangeldm_hip_ablate.so python tools/attn_timeline2.py [B L C]
For every wave of workgroup 0: s_memtime ticks since the workgroup's start at (affine, W', own tiles projected, barrier,
stabilisers, key loop done), then the start / end spread of all workgroups on the 100 MHz counter."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from rangeldm_amd import _lib  # noqa: E402

B, Lt, Cc = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (16, 1024, 128)
_lib.require_gpu()
L = _lib.lib()
L.rldm_debug_timestamps(None)
us = C.c_float()
_lib.check(L.rldm_bench_attention_qkv(B, Lt, Cc, 3, 5, C.byref(us), _lib.stream_ptr(torch.device("cuda"))), "bench")
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
L.rldm_debug_timestamps(buf)
print(f"B={B} L={Lt} C={Cc}: {us.value:.2f} us per launch (stamped build)")
t0 = min(buf[w * 8] for w in range(16) if buf[w * 8])
for w in range(16):
    v = [buf[w * 8 + i] for i in range(8)]
    if v[0]:
        print(f"  wave {w:2d}: start +{v[0] - t0:6d}  " + " ".join(f"{int(t - v[0]):7d}" if t else "      -" for t in v[1:]))
nb = 2048
bt = (C.c_ulonglong * (2 * nb))()
L.rldm_debug_block_times(bt, nb)
st = [bt[2 * i] for i in range(nb) if bt[2 * i]]
en = [bt[2 * i + 1] for i in range(nb) if bt[2 * i]]
if st:
    t0 = min(st)
    life = sorted((e - s) * 0.01 for s, e in zip(st, en))
    print(f"{len(st)} workgroups: starts spread {(max(st) - t0) * 0.01:.2f} us, ends {(min(en) - t0) * 0.01:.2f} .. "
          f"{(max(en) - t0) * 0.01:.2f} us; lifetime min / median / max {life[0]:.2f} / {life[len(life) // 2]:.2f} / {life[-1]:.2f} us")
