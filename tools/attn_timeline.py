#!/usr/bin/env python3
"""Phase stamps of the fused attention kernel inside one eager UNet forward (ABLATE build of the library):
RLDM_LIB=rangeldm_amd/librangeldm_hip_ablate.so RLDM_TS_ATTN_L=1024 python tools/attn_timeline.py
prints workgroup 0's s_memtime stamps (start, affine, W', projected, barrier, done) of the LAST launch with L == RLDM_TS_ATTN_L
and the start / end spread of all its workgroups on the 100 MHz counter."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rangeldm_amd import _lib  # noqa: E402

_lib.require_gpu()
_lib.lib().rldm_debug_timestamps(None)
p, unet, vae, _, _ = bench.build_models("RangeLDM", 20240310)
B = int(os.environ.get("B", "16"))
if os.environ.get("MODE", "forward") == "sampler":     # the production regime: captured 50-step chain + decode, real trajectory data
    from rangeldm_amd.pipelines import LDMPipelineRange
    from rangeldm_amd.schedulers import DDIMSchedulerHIP
    from rangeldm_amd.synth import latent_noise
    pipe = LDMPipelineRange(vae=vae, unet=unet, scheduler=DDIMSchedulerHIP(), pos_encoding=p["pos_encoding"])
    shape = (p["unet"].out_channels, *p["unet"].sample_size)
    xT = torch.from_numpy(np.stack([latent_noise(1, j, shape) for j in range(B)])).cuda()
    for _ in range(3):
        pipe(batch_size=B, num_inference_steps=50, latents=xT, output_type="torch")
else:
    x = torch.randn(B, p["unet"].in_channels, *p["unet"].sample_size, device="cuda")
    for _ in range(3):
        unet(x, 500)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
_lib.lib().rldm_debug_timestamps(buf)
v = [buf[i] for i in range(8) if buf[i]]
print("L =", os.environ.get("RLDM_TS_ATTN_L"), "workgroup 0 stamps (ticks since start):", [int(t - v[0]) for t in v])
nb = 2048
bt = (C.c_ulonglong * (2 * nb))()
_lib.lib().rldm_debug_block_times(bt, nb)
st = [bt[2 * i] for i in range(nb) if bt[2 * i]]
en = [bt[2 * i + 1] for i in range(nb) if bt[2 * i]]
t0 = min(st)
life = sorted((e - s) * 0.01 for s, e in zip(st, en))
print(f"{len(st)} workgroups: starts spread {(max(st) - t0) * 0.01:.2f} us, ends {(min(en) - t0) * 0.01:.2f} .. {(max(en) - t0) * 0.01:.2f} us; "
      f"lifetime min / median / max {life[0]:.2f} / {life[len(life) // 2]:.2f} / {life[-1]:.2f} us")
