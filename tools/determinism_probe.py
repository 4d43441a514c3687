#!/usr/bin/env python3
"""How far do two identically seeded trainers drift apart (eager vs eager, eager vs graphed)?  Fraction of parameters that differ by
more than lr / 2 after each step (Adam turns a sign flip of a noise-level gradient into a 2 lr difference)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rangeldm_amd.config import UNetConfig
from rangeldm_amd.params import unet_param_shapes
from rangeldm_amd.synth import synth_state_dict
from rangeldm_amd import training as TR
cfg = UNetConfig(sample_size=(32, 8), block_out_channels=(32, 32, 64, 64))
sd = synth_state_dict(unet_param_shapes(cfg), prefix="tr.")
lr = 1e-3
kw = dict(lr=lr, lr_warmup_steps=3, total_steps=40, use_ema=True)
a, b, c = (TR.UNetTrainer(cfg, sd, **kw) for _ in range(3))
g = torch.Generator().manual_seed(11)
for step in range(1, 10):
    x = torch.randn(2, 4, 32, 8, generator=g).cuda(); target = torch.randn(2, 4, 32, 8, generator=g).cuda()
    t = torch.randint(0, 1000, (2,), generator=g).cuda(); w = torch.rand(2, generator=g).cuda()
    a.train_step(x, t, target, w, pos_encoding=True)
    b.train_step(x, t, target, w, pos_encoding=True)
    c.train_step_graphed(x, t, target, w, pos_encoding=True)
    torch.cuda.synchronize()
    fb = float(((a.params - b.params).abs() > 0.5 * lr).float().mean())
    fc = float(((a.params - c.params).abs() > 0.5 * lr).float().mean())
    gb = float((a.grads - b.grads).abs().max())
    print(f"step {step}: eager-eager {fb:.4f}  eager-graphed {fc:.4f}  grad norm a/b/c {float(a.last_grad_norm)**0.5:.6f} {float(b.last_grad_norm)**0.5:.6f} {float(c.last_grad_norm)**0.5:.6f}")
