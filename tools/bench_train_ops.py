#!/usr/bin/env python3
"""Time single training ops (rangeldm_amd/csrc/train.hip) at the RangeLDM level shapes, batch 8: conv forward (= data
gradient), weight gradient, attention forward / backward.  One JSON line per case with TFLOP/s."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from rangeldm_amd import train_ops as T  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


B = 8
for name, W, H, Cin, N, taps in (("L0 128->128", 256, 16, 128, 128, 9), ("L0 up 256->128", 256, 16, 256, 128, 9),
                                 ("L1 128->128", 128, 8, 128, 128, 9), ("L1 up 256->128", 128, 8, 256, 128, 9),
                                 ("L1 upsample 256->256", 128, 8, 256, 256, 9),
                                 ("L2 256->256", 64, 4, 256, 256, 9), ("L2 up 512->256", 64, 4, 512, 256, 9),
                                 ("L3 256->256", 32, 2, 256, 256, 9), ("L3 up 512->256", 32, 2, 512, 256, 9),
                                 ("L0 1x1 256->128", 256, 16, 256, 128, 1), ("L2 1x1 256->256", 64, 4, 256, 256, 1)):
    x = torch.randn(B, W, H, Cin, device="cuda")
    w = torch.randn(N, Cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1, device="cuda") * 0.02
    wf, wt = T.pack_weights(w, taps)
    y = T.conv(x, wf, N, taps)
    dw = torch.zeros_like(w)
    fl = 2.0 * B * W * H * N * Cin * taps
    tc = timed(lambda: T.conv(x, wf, N, taps, out=y))
    tw = timed(lambda: T.wgrad(y, x, dw, taps))
    print(json.dumps({"op": name, "conv_us": round(tc * 1e6, 1), "conv_tflops": round(fl / tc / 1e12, 1),
                      "wgrad_us": round(tw * 1e6, 1), "wgrad_tflops": round(fl / tw / 1e12, 1)}))
for L, C in ((1024, 128), (256, 256), (64, 256)):
    q, k, v = (torch.randn(B, L, C, device="cuda") for _ in range(3))
    o, lse = T.attention_forward(q, k, v)
    tf = timed(lambda: T.attention_forward(q, k, v))
    tb = timed(lambda: T.attention_backward(q, k, v, o, o, lse))
    print(json.dumps({"op": f"attention L={L} C={C}", "fwd_us": round(tf * 1e6, 1), "bwd_us": round(tb * 1e6, 1)}))
