#!/usr/bin/env python3
"""Where a training step's GPU time goes, by op and shape: wraps the train_ops entry points with HIP events (eager launches,
batch 8, RangeLDM config, no VAE).  usage: python tools/train_breakdown.py [--top 40]"""
import argparse
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from rangeldm_amd import train_ops as T, training as TR  # noqa: E402
from rangeldm_amd.config import UNetConfig  # noqa: E402
from rangeldm_amd.params import unet_param_shapes  # noqa: E402
from rangeldm_amd.synth import synth_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--top", type=int, default=45)
a = ap.parse_args()
records = []


def wrap(name, keyfn):
    fn = getattr(T, name)

    def w(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        records.append((name, keyfn(*args, **kw), e0, e1))
        return out
    setattr(T, name, w)


shp = lambda t: "x".join(str(int(v)) for v in t.shape)      # noqa: E731
wrap("conv", lambda x, w, N, taps, stride=1, mode=0, **kw: f"{shp(x)}->{N} t{taps} s{stride} m{mode}")
wrap("wgrad", lambda dy, x, dw, taps, stride=1, mode=0: f"{shp(x)}->{dy.shape[3]} t{taps} s{stride} m{mode}")
wrap("colsum", lambda dy, **kw: shp(dy))
wrap("gn_forward", lambda x, *r, **kw: shp(x))
wrap("gn_backward", lambda x, *r, **kw: shp(x))
wrap("attention_qkv_forward", lambda qkv: shp(qkv))
wrap("attention_qkv_backward", lambda qkv, *r: shp(qkv))
wrap("add", lambda x, *r, **kw: shp(x))
wrap("copy_channels", lambda src, so, dst, do, n, **kw: f"{shp(src)}->{shp(dst)}")
wrap("linear_rows", lambda x, w, N, **kw: f"{shp(x)}->{N}")
wrap("linear_rows_wgrad", lambda dy, x, *r: f"{shp(x)}->{dy.shape[1]}")
wrap("sum2x2", lambda x: shp(x))
wrap("adamw", lambda p, *r, **kw: shp(p))
wrap("sqnorm", lambda g: shp(g))
# fused tape (round 5)
srcshp = lambda srcs: "+".join(shp(s_.t) for s_ in srcs)   # noqa: E731
wrap("wgrad_bias", lambda dy, x, dw, taps, stride=1, mode=0, **kw: f"{shp(x)}->{dy.shape[3]} t{taps} s{stride} m{mode}")
wrap("conv_fused", lambda srcs, w, N, taps, stride=1, mode=0, gn=None, want_stats=False, gsrcs=None, **kw:
     f"{srcshp(srcs)}->{N} t{taps} s{stride} m{mode}" + (" gn" if gn is not None else "") + (" st" if want_stats else "") + (" gnb" if gsrcs is not None else ""))
wrap("wgrad_fused", lambda dy, srcs, dw, taps, gn=None, **kw: f"{srcshp(srcs)}->{dy.shape[3]} t{taps}" + (" gn" if gn is not None else ""))
wrap("gn_backward_apply", lambda dz, srcs, *r, **kw: srcshp(srcs))
wrap("chan_stats", lambda x: shp(x))

cfg = UNetConfig()
tr = TR.UNetTrainer(cfg, synth_state_dict(unet_param_shapes(cfg)), use_ema=True)
_repack = tr.repack
x = torch.randn(8, 4, 256, 16).cuda(); tgt = torch.randn(8, 4, 256, 16).cuda(); t = torch.randint(0, 1000, (8,)).cuda()
for i in range(3):
    if i == 2:
        records.clear()
    tr.train_step(x, t, tgt, pos_encoding=True)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
byop = collections.defaultdict(lambda: [0, 0.0])
for name, key, e0, e1 in records:
    ms = e0.elapsed_time(e1)
    agg[(name, key)][0] += 1; agg[(name, key)][1] += ms
    byop[name][0] += 1; byop[name][1] += ms
tot = sum(v[1] for v in byop.values())
print(f"total (event-timed ops, eager: includes launch gaps) {tot:.2f} ms")
for name, (n, ms) in sorted(byop.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:26s} {n:4d} calls {ms:7.3f} ms")
print()
for (name, key), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"{name:24s} {key:38s} {n:3d} x {ms / n * 1e3:7.1f} us = {ms:6.3f} ms")
