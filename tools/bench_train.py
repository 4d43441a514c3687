#!/usr/bin/env python3
"""Training throughput of BASELINE config 5 (RangeLDM KITTI-360 unconditional, bf16 operands, AdamW, batch 8 per GPU,
data-parallel): samples/sec = global batch / step time, same barrier + max-over-ranks timing as bench.py.

    python tools/bench_train.py [--batch 8] [--steps 5] [--warmup 2] [--no-vae]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py --gpus N ...

One step = the loop body of ldm/train_unconditional.py:479-556 on synthetic range images resident in HBM:
VAE encode + sample -> add_noise -> pos-encoding -> UNet forward -> MSE -> backward -> (RCCL gradient all-reduce) ->
clip -> AdamW -> EMA -> operand repack.  Algorithmic work per sample: 3 x 34.07 GFLOP (UNet fwd + bwd) + 75.7 GFLOP (VAE
encode) = 177.9 GFLOP.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--overlap-vae", action="store_true", help="encode batch i + 1 on a second stream while batch i trains (measured: 4 % slower than in-step)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of replaying the captured step graphs")
    ap.add_argument("--prio", action="store_true", help="with --overlap-vae: the training step on a HIGH-priority stream, the encode on a default-priority one")
    ap.add_argument("--seed", type=int, default=20240310)
    a = ap.parse_args()
    from rangeldm_amd import distributed as D
    from rangeldm_amd.config import PRESETS
    from rangeldm_amd.params import unet_param_shapes, vae_param_shapes
    from rangeldm_amd.schedulers import DDPMSchedulerHIP
    from rangeldm_amd.synth import synth_state_dict, normal
    from rangeldm_amd.training import UNetTrainer, training_step, encode_ahead
    from rangeldm_amd.vae import AutoencoderKLHIP
    rank, world, local = D.init_from_env("nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    p = PRESETS["RangeLDM"]
    tr = UNetTrainer(p["unet"], synth_state_dict(unet_param_shapes(p["unet"]), seed=a.seed), device=dev)
    vae = None
    if not a.no_vae:
        vae = AutoencoderKLHIP(p["vae"])
        vae.load_state_dict(synth_state_dict(vae_param_shapes(p["vae"]), seed=a.seed, prefix="vae."))
    sched = DDPMSchedulerHIP()
    B = a.batch
    if not a.eager:
        a.warmup = max(a.warmup, 2)                    # (step 1 runs eagerly and sizes the scratch buffers, step 2 captures)
    n_iter = a.warmup + a.steps
    gen = torch.Generator().manual_seed(a.seed + rank)
    shape = (B, 2, 1024, 64) if vae is not None else (B, 4, 256, 16)
    imgs = [torch.from_numpy(normal(a.seed, f"train/{rank}/{i}", shape)).mul_(0.5).to(dev) for i in range(n_iter)]
    losses = []

    side = torch.cuda.Stream(dev) if (vae is not None and a.overlap_vae) else None
    ahead = {}

    def one(i):
        if side is None:
            losses.append(training_step(tr, vae, sched, imgs[i], generator=gen, pos_encoding=True, graphed=not a.eager))
            return
        # input pipelining: batch i + 1 is encoded on a second stream while batch i trains (every step still encodes one batch)
        if i not in ahead:
            ahead[i] = encode_ahead(vae, imgs[i], side, generator=gen)
        lat, ev = ahead.pop(i)
        torch.cuda.current_stream(dev).wait_event(ev)
        if i + 1 < n_iter:
            ahead[i + 1] = encode_ahead(vae, imgs[i + 1], side, generator=gen)
        losses.append(training_step(tr, None, sched, imgs[i], generator=gen, pos_encoding=True, graphed=not a.eager, latents=lat))

    import contextlib
    main_ctx = contextlib.nullcontext()
    if a.prio:
        lo, hi = torch.cuda.Stream.priority_range()
        main_ctx = torch.cuda.stream(torch.cuda.Stream(dev, priority=hi))
    with main_ctx:
        for i in range(a.warmup):
            one(i)
        torch.cuda.synchronize(); D.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.warmup, n_iter):
            one(i)
        torch.cuda.synchronize(); D.barrier(); torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        sps = world * B * a.steps / dt
        gflop = 3 * 34.071 + (75.73 if vae is not None else 0.0)
        print(json.dumps({"metric": "training samples/sec, RangeLDM KITTI-360 unconditional, bf16 operands, AdamW",
                          "value": sps, "unit": "samples/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "dtype": "bf16",
                          "data": "synthetic", "config": {"workload": f"train_unconditional step, batch {B} per GPU, "
                                                          f"{'VAE encode + ' if vae is not None else ''}UNet fwd+bwd, AdamW, EMA",
                                                          "launch": "eager" if a.eager else "captured HIP graphs",
                                                          "vae_encode": "in step" if side is None else "one batch ahead on a second stream",
                                                          "global_batch": B * world},
                          "gflop_per_sample": gflop, "end_to_end_tflops": sps * gflop / 1e3,
                          "loss_first_last": [float(losses[0]), float(losses[-1])]}))
    D.barrier()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
