#!/usr/bin/env python3
"""Two data-parallel ranks on ONE GPU (backend gloo on device tensors: RCCL refuses two ranks per device) through the captured,
bucket-cut training step: both ranks must end with identical parameters, equal to one process trained on the concatenated batch.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_probe.py [--eager] [--full]
--full: the RangeLDM-size UNet on 256 x 16 latents (the fused blocks, the weight-gradient riders and their flush at every bucket cut
only exist at this size), 3 steps."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from rangeldm_amd.config import UNetConfig
from rangeldm_amd.params import unet_param_shapes
from rangeldm_amd.synth import synth_state_dict
from rangeldm_amd import training as TR

eager = "--eager" in sys.argv
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
full = "--full" in sys.argv
cfg = UNetConfig() if full else UNetConfig(sample_size=(32, 8), block_out_channels=(32, 32, 64, 64))
W, H = cfg.sample_size
sd = synth_state_dict(unet_param_shapes(cfg), prefix="tr.")
lr = 1e-4 if full else 1e-3
kw = dict(lr=lr, lr_warmup_steps=2, total_steps=50, use_ema=True, bucket_mb=32 if full else 1)
tr = TR.UNetTrainer(cfg, sd, **kw)
ref = TR.UNetTrainer(cfg, sd, **kw) if rank == 0 else None
g = torch.Generator().manual_seed(7)
B = 2
for step in range(1, 5 if full else 7):
    x = torch.randn(world * B, 5, W, H, generator=g).cuda()
    tgt = torch.randn(world * B, 4, W, H, generator=g).cuda()
    t = torch.randint(0, 1000, (world * B,), generator=g).cuda()
    sl = slice(rank * B, (rank + 1) * B)
    if eager:
        tr.train_step(x[sl].contiguous(), t[sl].contiguous(), tgt[sl].contiguous())
    else:
        tr.train_step_graphed(x[sl].contiguous(), t[sl].contiguous(), tgt[sl].contiguous())
    if ref is not None:                                  # one process, the whole batch, no collectives
        pred = ref.forward(x, t)
        from rangeldm_amd import train_ops as T
        loss, dpred = T.mse(pred, tgt)
        ref.backward(dpred, reduce=False)
        ref.optimizer_step(1)
torch.cuda.synchronize()
mine = tr.params.detach().cpu()
both = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
if rank == 0:
    d_ranks = float((both[0] - both[1]).abs().max())
    d_ref = (both[0] - ref.params.cpu()).abs()
    frac = float((d_ref > 0.5 * lr).float().mean())
    segs = None if eager else [a for _, a in next(iter(tr._graphs.values()))["segments"]]
    print(f"dp_probe ({'eager' if eager else 'graphed'}): max |rank0 - rank1| = {d_ranks:.3e}; fraction of parameters off the "
          f"single-process run by > lr/2: {frac:.4f}; segments {segs}")
    assert d_ranks == 0.0, "ranks diverged"
    assert frac < (0.05 if full else 0.02), frac
    if full:
        assert tr.last_forward_fused_levels == [True, True, False, False] and len(segs) >= 3, (tr.last_forward_fused_levels, segs)
dist.barrier()
dist.destroy_process_group()
