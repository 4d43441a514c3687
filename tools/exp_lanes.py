#!/usr/bin/env python3
"""Experiment (round 6): do independent micro-batch lanes of the training step overlap on separate HIP streams?

    python tools/exp_lanes.py [--steps 10]

Leg A: one trainer, batch B in {8, 4, 2, 1}: ms per step (how much of the step is launch-latency floor).
Leg B: L trainers of batch 8 / L, each replaying its captured step graphs on its own stream, all L in flight: ms per L-step round.
Timing only: the lanes' trainers are separate objects (separate parameters), nothing is compared.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--seed", type=int, default=20240310)
    a = ap.parse_args()
    from rangeldm_amd.config import PRESETS
    from rangeldm_amd.params import unet_param_shapes
    from rangeldm_amd.synth import synth_state_dict, normal
    from rangeldm_amd.training import UNetTrainer
    dev = torch.device("cuda", 0)
    p = PRESETS["RangeLDM"]
    sd = synth_state_dict(unet_param_shapes(p["unet"]), seed=a.seed)

    def data(B, tag):
        x = torch.from_numpy(normal(a.seed, f"lanes/x/{tag}", (B, 4, 256, 16))).to(dev)
        n = torch.from_numpy(normal(a.seed, f"lanes/n/{tag}", (B, 4, 256, 16))).to(dev)
        t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(7)).to(dev)
        return x, t, n

    out = {}
    for B in (8, 4, 2, 1):
        tr = UNetTrainer(p["unet"], sd, device=dev)
        x, t, n = data(B, B)
        for _ in range(3):
            tr.train_step_graphed(x, t, n, None, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            tr.train_step_graphed(x, t, n, None, True)
        torch.cuda.synchronize()
        out[f"single_B{B}_ms"] = round((time.perf_counter() - t0) / a.steps * 1e3, 3)
        del tr
        torch.cuda.empty_cache()
    for L in (2, 4):
        B = 8 // L
        trs = [UNetTrainer(p["unet"], sd, device=dev) for _ in range(L)]
        streams = [torch.cuda.Stream(dev) for _ in range(L)]
        ds = [data(B, f"{L}/{i}") for i in range(L)]
        for i in range(L):
            with torch.cuda.stream(streams[i]):
                for _ in range(3):
                    trs[i].train_step_graphed(*ds[i][:3], None, True)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            for i in range(L):
                with torch.cuda.stream(streams[i]):
                    trs[i].train_step_graphed(*ds[i][:3], None, True)
        torch.cuda.synchronize()
        out[f"lanes{L}_B{B}_ms_per_round"] = round((time.perf_counter() - t0) / a.steps * 1e3, 3)
        del trs
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
