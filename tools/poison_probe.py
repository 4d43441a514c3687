#!/usr/bin/env python3
"""Uninitialised-read hunt for the training step: every torch.empty / torch.empty_like comes back filled with NaN (fp) so that an
op which reads memory nobody wrote turns its output into NaN; prints the first op whose output holds a NaN.  Ops that legitimately
overwrite their whole output are unaffected."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_empty, _empty_like = torch.empty, torch.empty_like
def p_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t
def p_empty_like(x, *a, **k):
    t = _empty_like(x, *a, **k)
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
    return t
torch.empty, torch.empty_like = p_empty, p_empty_like
from rangeldm_amd import training as TR, train_ops as T
from rangeldm_amd.config import UNetConfig
from rangeldm_amd.params import unet_param_shapes
from rangeldm_amd.synth import synth_state_dict
small = len(sys.argv) < 2 or sys.argv[1] != "full"
cfg = UNetConfig(sample_size=(32, 8), block_out_channels=(32, 32, 64, 64)) if small else UNetConfig()
sd = synth_state_dict(unet_param_shapes(cfg), prefix="tr.")
B = 2
x = torch.randn(B, 5, *cfg.sample_size, generator=torch.Generator().manual_seed(1)).cuda()
target = torch.randn(B, 4, *cfg.sample_size, generator=torch.Generator().manual_seed(2)).cuda()
t = torch.tensor([5, 900]).cuda()
names = [n for n in dir(T) if callable(getattr(T, n)) and not n.startswith("_") and n not in ("empty", "conv_desc", "out_size", "set_zero_arena", "ZeroArena", "pack_weights")]
log = []
def wrap(n):
    f = getattr(T, n)
    def g(*a, **k):
        r = f(*a, **k)
        outs = r if isinstance(r, (tuple, list)) else (r,)
        for o in outs:
            if torch.is_tensor(o) and o.is_floating_point() and bool(torch.isnan(o).any()):
                log.append((n, tuple(o.shape), int(torch.isnan(o).sum())))
        for key in ("out", "dx", "rows", "total", "dw", "dgamma", "dbeta", "dbias"):
            o = k.get(key)
            if torch.is_tensor(o) and bool(torch.isnan(o).any()):
                log.append((n + "[" + key + "]", tuple(o.shape), int(torch.isnan(o).sum())))
        return r
    return g
for n in names:
    if n[0].islower():
        setattr(T, n, wrap(n))
tr = TR.UNetTrainer(cfg, sd, use_ema=False, bucket_mb=1)
pred = tr.forward(x, t)
print("forward: first NaN-producing ops:", log[:5])
log.clear()
loss, dpred = T.mse(pred, target)
tr.backward(dpred, reduce=False)
torch.cuda.synchronize()
print("backward: first NaN-producing ops:", log[:8])
print("NaNs in the flat gradient buffer:", int(torch.isnan(tr.grads).sum()), "of", tr.grads.numel())
bad = [n for n in tr.names if bool(torch.isnan(tr.grads[tr.offsets[n]:tr.offsets[n] + tr.sizes[n]]).any())]
print("parameters with NaN gradients:", bad[:10])
