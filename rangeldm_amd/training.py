"""UNet training step on MI355X: the counterpart of ldm/train_unconditional.py:466-558.

`UNetTrainer` owns flat fp32 buffers (parameters, gradients, AdamW moments, EMA copy; state-dict views into them) and
drives the op-level HIP kernels of rangeldm_amd/csrc/train.hip through a host-side tape: `forward` records one backward
closure per op (conv / linear, GroupNorm(+SiLU), head_dim-8 attention, concat, nearest-x2, stride-2), `backward` replays
them in reverse.  Data parallelism = one process per GPU; gradients are averaged with RCCL all-reduce over a few large
buckets of the flat gradient buffer, each launched as soon as backward has finished the bucket's parameters
(bucket boundaries: rangeldm_amd.training.plan_buckets).  No torch autograd, no torch compute ops on the path (torch
allocates and zeroes buffers and runs the collectives); no CPU fallback.

Parity: tests/test_training.py compares every parameter gradient of a small UNet, and the parameters after optimizer
steps, with torch autograd / torch.optim.AdamW on the oracle (oracle/unet.py).
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from . import train_ops as T
from .config import UNetConfig
from .params import unet_param_shapes


def plan_buckets(sizes, target_elems):
    """Contiguous buckets over the flat gradient buffer, in parameter order: [(first_param, end_param, offset, numel)].
    Backward finishes parameters in REVERSE order, so the last bucket is reduced first."""
    out, start, off, acc, pos = [], 0, 0, 0, 0
    for i, n in enumerate(sizes):
        acc += n
        pos += n
        if acc >= target_elems or i == len(sizes) - 1:
            out.append((start, i + 1, off, acc))
            start, off, acc = i + 1, pos, 0
    return out


def plan_parameter_order(names, shapes):
    """Order of the parameters inside the flat buffers (state-dict views do not care) and the fused layer groups it makes
    possible: every ResnetBlock2D.time_emb_proj reads the same silu(temb), so their weights laid end to end are ONE
    Linear(512 -> sum of channels) (one launch instead of 22 forward, 22 backward); to_q / to_k / to_v of an Attention read
    the same normalised input: one 1x1 conv with 3C outputs.  Returns (names, {key: {"weights": [...], "biases": [...]}});
    a group's weights are consecutive in `names`, so are its biases."""
    fused = OrderedDict()
    tw = [n for n in names if n.endswith(".time_emb_proj.weight")]
    if len(tw) > 1 and len({shapes[n][1] for n in tw}) == 1:
        fused["time_emb_proj_all"] = {"weights": tw, "biases": [n[:-len("weight")] + "bias" for n in tw]}
    for n in names:
        if n.endswith(".to_q.weight"):
            pre = n[:-len("to_q.weight")]
            trio = [pre + f"to_{c}.weight" for c in "qkv"]
            if all(t in shapes and shapes[t] == shapes[n] for t in trio):
                fused[pre + "to_qkv"] = {"weights": trio, "biases": [pre + f"to_{c}.bias" for c in "qkv"]}
    members = {m for grp in fused.values() for m in grp["weights"] + grp["biases"]}
    out = []
    temb_done = "time_emb_proj_all" not in fused
    for n in names:
        if not temb_done and n.startswith(("down_blocks", "mid_block", "up_blocks")):
            out += fused["time_emb_proj_all"]["weights"] + fused["time_emb_proj_all"]["biases"]
            temb_done = True
        if n in members:
            if n.endswith(".to_q.weight"):
                grp = fused[n[:-len("to_q.weight")] + "to_qkv"]
                out += grp["weights"] + grp["biases"]
            continue
        out.append(n)
    assert sorted(out) == sorted(names)
    return out, fused


def cosine_lr(step, base_lr, warmup, total):
    """diffusers get_scheduler("cosine", num_warmup_steps, num_training_steps) (ldm/train_unconditional.py:394-399)."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    progress = (step - warmup) / max(1, total - warmup)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


def ema_decay(optimization_step, max_decay=0.9999, inv_gamma=1.0, power=0.75, min_decay=0.0):
    """diffusers EMAModel.get_decay with use_ema_warmup=True (ldm/train_unconditional.py:320-329)."""
    step = max(0, optimization_step - 1)
    if step <= 0:
        return 0.0
    return max(min(1.0 - (1.0 + step / inv_gamma) ** -power, max_decay), min_decay)


def snr_weights(alphas_cumprod, timesteps, snr_gamma, v_prediction=False):
    """min(SNR, gamma) / SNR (ldm/train_unconditional.py:529-538); v_prediction: "add one to SNR values before we divide by them"
    (:532-534 -- the +1 goes into BOTH the min and the divisor, as the reference writes it)."""
    ac = alphas_cumprod[timesteps.cpu()].double()
    snr = ac / (1.0 - ac)
    if v_prediction:
        snr = snr + 1.0
    return (torch.minimum(snr, torch.full_like(snr, snr_gamma)) / snr).float()


def fused_tape_supported(cfg, fused_groups, B, W, H):
    """Can the fused blocks (rldm_train_conv_fused / _wgrad_fused) run EVERY layer of this UNet at this input size?  Pure shape
    logic: pixel tiles of 64 inside one image at every level, 64-channel chunks inside one source of a concatenation (and <= 768
    input channels), the all-taps weight-gradient kernel's geometry (2 - 16 beams, a power of two; azimuth a multiple of 8),
    batch <= 16 (its per-image coefficient table), and the fused parameter groups (all time_emb_proj layers; to_q / to_k / to_v of
    every attention block).  `fused_groups`: the keys of plan_parameter_order's second result."""
    if B > 16 or "time_emb_proj_all" not in fused_groups:
        return False
    if cfg.norm_num_groups > 64 or any(c % 64 or c % cfg.norm_num_groups or c > 384 for c in cfg.block_out_channels):
        return False
    for lvl in range(len(cfg.block_out_channels)):
        w, h = W >> lvl, H >> lvl
        if (w << lvl) != W or (h << lvl) != H or (w * h) % 64 or h < 2 or h > 16 or (h & (h - 1)) or w % 8:
            return False
    attn = [f"down_blocks.{i}.attentions.{j}" for i, bt in enumerate(cfg.down_block_types) if bt == "AttnDownBlock2D"
            for j in range(cfg.layers_per_block)]
    attn += [f"up_blocks.{i}.attentions.{j}" for i, bt in enumerate(cfg.up_block_types) if bt == "AttnUpBlock2D"
             for j in range(cfg.layers_per_block + 1)]
    if cfg.add_attention:
        attn.append("mid_block.attentions.0")
    return all((a + ".to_qkv") in fused_groups for a in attn)


def fused_levels(W, H, num_levels, min_pixels):
    """Per level: do its blocks run fused?  (from `min_pixels` pixels per image on: measured per shape)"""
    return [(W >> l) * (H >> l) >= min_pixels for l in range(num_levels)]


class UNetTrainer:
    def __init__(self, config, state_dict, device="cuda", lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8,
                 max_grad_norm=1.0, use_ema=True, ema_max_decay=0.9999, ema_inv_gamma=1.0, ema_power=0.75,
                 lr_warmup_steps=500, total_steps=100000, bucket_mb=32, gradient_accumulation_steps=1):
        self.cfg = config if isinstance(config, UNetConfig) else UNetConfig(**config)
        if not self.cfg.flip_sin_to_cos or self.cfg.freq_shift != 0:
            raise NotImplementedError("the training step implements UNet2DModel's default time embedding "
                                      "(flip_sin_to_cos=True, freq_shift=0) only")
        _lib.require_gpu()
        self.device = torch.device(device)
        self.shapes = unet_param_shapes(self.cfg)
        self.names, self.fused = plan_parameter_order(list(self.shapes), self.shapes)
        sizes = [int(np.prod(self.shapes[n])) for n in self.names]
        self.offsets = dict(zip(self.names, np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()))
        self.sizes = dict(zip(self.names, sizes))
        self.numel = int(sum(sizes))
        z = lambda: torch.zeros(self.numel, dtype=torch.float32, device=self.device)      # noqa: E731
        self.params, self.grads, self.exp_avg, self.exp_avg_sq = z(), z(), z(), z()
        self.ema = z() if use_ema else None
        host = np.empty(self.numel, np.float32)
        for n in self.names:
            a = state_dict[n]
            a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
            if tuple(a.shape) != tuple(self.shapes[n]):
                raise RuntimeError(f"size mismatch for {n}: {tuple(a.shape)} vs {tuple(self.shapes[n])}")
            host[self.offsets[n]:self.offsets[n] + self.sizes[n]] = a.reshape(-1)
        self.params.copy_(torch.from_numpy(host))
        if self.ema is not None:
            self.ema.copy_(self.params)
        self.p = {n: self._view(self.params, n) for n in self.names}
        self.g = {n: self._view(self.grads, n) for n in self.names}
        # bf16 operand copies of every conv / linear weight: forward [N][taps][Cin], data gradient [Cin][taps][N].
        # layers: key -> (N, Cin, taps, offset of the fp32 weight in the flat buffer).  A fused group (all time_emb_proj
        # layers; to_q / to_k / to_v of an attention block) is ONE layer over its members' contiguous parameters.
        self.layers, self.wf, self.wt = OrderedDict(), {}, {}
        members = {m for grp in self.fused.values() for m in grp["weights"] + grp["biases"]}
        for n in self.names:
            if n.endswith(".weight") and len(self.shapes[n]) >= 2 and n not in members:
                N, Cin = self.shapes[n][:2]
                taps = 9 if len(self.shapes[n]) == 4 and self.shapes[n][2] == 3 else 1
                self.layers[n] = (N, Cin, taps, self.offsets[n], n not in ("conv_in.weight", "time_embedding.linear_1.weight"))
        for key, grp in self.fused.items():
            N = sum(self.shapes[m][0] for m in grp["weights"])
            Cin = self.shapes[grp["weights"][0]][1]
            ow, ob = self.offsets[grp["weights"][0]], self.offsets[grp["biases"][0]]
            self.layers[key + ".weight"] = (N, Cin, 1, ow, True)
            self.p[key + ".bias"] = self.params[ob:ob + N]
            self.g[key + ".bias"] = self.grads[ob:ob + N]
            self.g[key + ".weight"] = self.grads[ow:ow + N * Cin].view(N, Cin)
        for n, (N, Cin, taps, _, need_t) in self.layers.items():
            self.wf[n] = torch.empty((N, taps, (Cin + 15) // 16 * 16), dtype=torch.bfloat16, device=self.device)
            self.wt[n] = torch.empty((Cin, taps, (N + 15) // 16 * 16), dtype=torch.bfloat16, device=self.device) if need_t else None
        self.repack()
        self.hp = dict(lr=lr, betas=betas, weight_decay=weight_decay, eps=eps, max_grad_norm=max_grad_norm,
                       ema_max_decay=ema_max_decay, ema_inv_gamma=ema_inv_gamma, ema_power=ema_power,
                       lr_warmup_steps=lr_warmup_steps, total_steps=total_steps)
        self.global_step = 0
        # `gradient_accumulation_steps` of the reference's yaml (accelerator.accumulate(model), ldm/train_unconditional.py:
        # 466,509): the optimizer runs on every k-th train_step over the mean of the k micro-batch gradients
        self.gradient_accumulation_steps = int(gradient_accumulation_steps)
        self._micro = 0
        self.buckets = plan_buckets([self.sizes[n] for n in self.names], bucket_mb * (1 << 20) // 4)
        self._tape, self._grad, self._keep = [], {}, []
        self._rows, self._row_parent = {}, {}
        self._arena = T.ZeroArena(self.device, 256 << 20)
        self._pending, self._ready = [], None
        self.last_grad_norm = None
        # captured step graphs (train_step_graphed): device-side step counter + per-step optimizer scalars
        self._graphs, self._on_bucket = {}, None
        self._step_dev = torch.zeros((), dtype=torch.int64, device=self.device)
        self._step_dev_mirror = 0
        self._dyn = torch.zeros(4, dtype=torch.float32, device=self.device)
        # fused tape (round 5): GroupNorm folded into its producers / consumers, concatenations read in place
        # (RLDM_TRAIN_FUSED=0: the op-per-layer tape, kept as the cross-check)
        self.fused_tape = os.environ.get("RLDM_TRAIN_FUSED", "1") != "0"
        self.fused_min_pixels = int(os.environ.get("RLDM_TRAIN_FUSED_MINPX", "1024"))
        # the reduction of a weight gradient's partial tiles rides on the layer's data-gradient launch (RLDM_TR_DEFER_REDUCE=0: own launch)
        self.defer_reduce = os.environ.get("RLDM_TR_DEFER_REDUCE", "1") != "0"
        # (round 6) the weight gradients of the step are queued and run as a few grouped launches (rldm_train_wgrad_group: nothing in
        # backward waits for a weight gradient, and ~85 launches of them each sat at its launch floor); RLDM_TR_WGRAD_GROUP=0: launched
        # where the tape reaches them.  _wg_keep: the queued launches' operands (dy, inputs, statistics), alive until the flush
        self.wgrad_group = os.environ.get("RLDM_TR_WGRAD_GROUP", "1") != "0"
        self._wg_on, self._wg_keep = False, []
        self._cs = {}

    # ---- parameters -------------------------------------------------------------------------------------------
    def _view(self, flat, n):
        return flat[self.offsets[n]:self.offsets[n] + self.sizes[n]].view(self.shapes[n])

    def repack(self, tiled=True):
        """Refresh the bf16 operand copies of every conv / linear weight from the fp32 masters: one launch (tiled: a 64 x 64
        tile transpose per workgroup; False: the element-wise kernel, kept as the cross-check)."""
        import ctypes as C
        if getattr(self, "_pack_table", None) is None:
            tables = []
            for per_tile in (False, True):
                descs = (_lib.PackDescC * len(self.wf))()
                first = 0
                for i, (n, wf) in enumerate(self.wf.items()):
                    N, Cin, _, offset, _ = self.layers[n]
                    wt = self.wt[n]
                    d = descs[i]
                    d.first, d.param_offset = first, offset
                    d.w_forward = wf.data_ptr()
                    d.w_transposed = wt.data_ptr() if wt is not None else None
                    d.N, d.Cin, d.taps = N, Cin, wf.shape[1]
                    first += (((N + 63) // 64) * ((Cin + 63) // 64)) if per_tile else max(wf.numel(), wt.numel() if wt is not None else 0)
                tables.append((torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.device), first))
            self._pack_table = tables
        table, total = self._pack_table[1 if tiled else 0]
        fn = _lib.lib().rldm_train_pack_weights_tiled if tiled else _lib.lib().rldm_train_pack_weights_all
        _lib.check(fn(C.c_void_p(self.params.data_ptr()), C.c_void_p(table.data_ptr()), len(self.wf), total,
                      _lib.stream_ptr(self.device)), "rldm_train_pack_weights")

    def state_dict(self, ema=False):
        flat = (self.ema if ema else self.params).cpu()
        return OrderedDict((n, flat[self.offsets[n]:self.offsets[n] + self.sizes[n]].view(self.shapes[n]).clone())
                           for n in self.names)

    def save_pretrained(self, output_dir):
        """`unet/` (+ `unet_ema/`) in the layout of the reference's save hook (ldm/train_unconditional.py:148-153)."""
        import os
        from .checkpoint import save_model_dir, unet_config_to_diffusers
        save_model_dir(os.path.join(output_dir, "unet"), unet_config_to_diffusers(self.cfg), self.state_dict())
        if self.ema is not None:
            save_model_dir(os.path.join(output_dir, "unet_ema"), unet_config_to_diffusers(self.cfg), self.state_dict(ema=True))

    def save_state(self, path):
        """`accelerator.save_state(checkpoint-N)` counterpart (ldm/train_unconditional.py:560-584): everything a resumed run
        needs besides the weights -- AdamW moments, EMA copy, step counter (lr schedule, bias corrections, EMA warm-up)."""
        # inside an accumulation window the partially summed gradient buffer is part of the state (the next optimizer step divides by
        # the window length whatever the buffer holds)
        torch.save({"params": self.params.cpu(), "ema": None if self.ema is None else self.ema.cpu(),
                    "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(), "global_step": self.global_step,
                    "micro": self._micro, "grads": self.grads.cpu() if self._micro else None,
                    "names": list(self.names), "hp": {k: (v if isinstance(v, (int, float, str, bool, type(None))) else list(v))
                                                      for k, v in self.hp.items()}}, path)

    def load_state(self, path):
        """`accelerator.load_state` (resume_from_checkpoint, ldm/train_unconditional.py:449-463): restores the buffers IN PLACE
        (captured step graphs keep pointing at them) and resynchronises the device-side step counter."""
        st = torch.load(path, map_location="cpu", weights_only=True)       # tensors, lists, dicts, numbers: nothing to unpickle
        if list(st["names"]) != list(self.names):
            raise RuntimeError("training state was saved for a different parameter layout")
        self.params.copy_(st["params"])
        self.exp_avg.copy_(st["exp_avg"])
        self.exp_avg_sq.copy_(st["exp_avg_sq"])
        if self.ema is not None:
            if st["ema"] is None:
                raise RuntimeError("training state holds no EMA copy")
            self.ema.copy_(st["ema"])
        self.global_step = int(st["global_step"])
        self._micro = int(st.get("micro", 0))
        self.grads.zero_()
        if self._micro:
            if st.get("grads") is None:                  # (a state written before the gradient buffer was saved: restart the window)
                self._micro = 0
            else:
                self.grads.copy_(st["grads"])
        self._step_dev.fill_(self.global_step)
        self._step_dev_mirror = self.global_step
        self.repack()

    # ---- tape -------------------------------------------------------------------------------------------------
    def _acc(self, t, g, owned):
        """Add gradient g to tensor t's slot; `owned`: g is a fresh buffer nobody else reads."""
        if t is None:
            return
        k = id(t)
        if k not in self._grad:
            self._grad[k] = (g, owned)
        else:
            cur, cur_owned = self._grad[k]
            self._grad[k] = (T.add(cur, g, out=cur), True) if cur_owned else (T.add(cur, g), True)

    def _pop(self, t):
        g = self._grad.pop(id(t), None)
        self._popped_owned = bool(g is not None and g[1])
        return None if g is None else g[0]

    def _slot(self, t):
        """The gradient buffer tensor t already has, if this tape owns it: kernels that can accumulate write into it directly
        (no separate add launch)."""
        g = self._grad.get(id(t))
        return g[0] if g is not None and g[1] else None

    def _pin(self, *tensors):
        """Operands of a (possibly queued) weight-gradient launch: kept alive until the queue is flushed."""
        if self._wg_on:
            self._wg_keep.extend(t for t in tensors if t is not None)

    def _wg_flush(self):
        """Run the queued weight gradients (their dw / bias / time-embedding-row outputs are final behind this)."""
        if self._wg_on:
            T.wgrad_group_flush()
            self._wg_keep.clear()

    def _done(self, *names):
        """The gradients of these parameters are final: launch the all-reduce of every bucket that just completed."""
        if self._ready is None:
            return
        for n in names:
            i = self._pidx[n]
            for b, (lo, hi, off, cnt) in enumerate(self.buckets):
                if lo <= i < hi:
                    self._ready[b] -= 1
                    if self._ready[b] == 0:
                        self._wg_flush()                         # (the bucket's weight gradients may still be queued)
                        T.flush_reduce()                         # (the last weight gradient's reduction may still be waiting for a conv to ride on)
                        if self._on_bucket is not None:          # stream capture: cut the graph here, reduce at replay
                            self._on_bucket(b)
                        else:
                            self._reduce_bucket(b)

    def _reduce_bucket(self, b):
        _, _, off, cnt = self.buckets[b]
        self._pending.append(self._launch_reduce(off, cnt, self._reduce_op))

    def _launch_reduce(self, off, cnt, op):
        """All-reduce of one gradient bucket, asynchronous to the stream that goes on with backward; returns an object whose
        wait() makes the current stream wait for it.  RLDM_COLLECTIVE=cabi: rldm_allreduce_grads (RCCL through the C ABI,
        include/rangeldm_hip.h) on a side stream; default: torch.distributed (backend "nccl" is the same RCCL)."""
        from . import distributed as D
        comm = D.cabi_communicator()
        if comm is None:
            return torch.distributed.all_reduce(self.grads[off:off + cnt], op=op, async_op=True)
        if getattr(self, "_comm_stream", None) is None:
            self._comm_stream = torch.cuda.Stream(self.device)
        side = self._comm_stream
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            comm.all_reduce_grads(self.grads[off:off + cnt], average=(op == torch.distributed.ReduceOp.AVG))

        class _Work:
            def wait(_self):
                torch.cuda.current_stream(self.device).wait_stream(side)
        return _Work()

    # ---- ops --------------------------------------------------------------------------------------------------
    def _conv(self, x, name, stride=1, mode=0, rowadd=None, res=None, need_dx=True, done=None, stats=False):
        w = name + ".weight"
        N, _, taps = self.layers[w][:3]
        done = done or (w, name + ".bias")
        if stats and rowadd is None and res is None and T.conv_fused_ok([T.Src(x)], N, taps, stride, mode, want_stats=True):
            y, cs_y = T.conv_fused([T.Src(x)], self.wf[w], N, taps, stride, mode, bias=self.p[name + ".bias"], want_stats=True)
            self._cs_set(y, cs_y)
        else:
            y = T.conv(x, self.wf[w], N, taps, stride, mode, bias=self.p[name + ".bias"], rowadd=rowadd, res=res)

        def bwd():
            dy = self._pop(y)
            dy_owned = self._popped_owned
            drow = None
            if rowadd is not None:
                parent = self._row_parent.get(id(rowadd))
                if parent is None:
                    drow = T.empty(rowadd.shape, rowadd)
                else:                                   # a slice of the fused time_emb_proj output: write its slice of the gradient
                    rows, off = parent
                    if id(rows) not in self._grad:       # zeroed once for all 22 slices (they accumulate into it)
                        self._grad[id(rows)] = (torch.zeros(rows.shape, dtype=torch.float32, device=rows.device), True)
                    drow = self._grad[id(rows)][0][:, off:off + rowadd.shape[1]]
            T.wgrad_bias(dy, x, self.g[w], taps, stride, mode, rows=drow, total=self.g[name + ".bias"],
                         rows_accumulate=rowadd is not None and self._row_parent.get(id(rowadd)) is not None)
            self._pin(dy, x, drow)
            self._done(*done)
            if rowadd is not None and self._row_parent.get(id(rowadd)) is None:
                self._acc(rowadd, drow, True)
            if res is not None:
                # (everything below that reads dy is enqueued before anyone adds to it -- unless the weight gradient above is
                #  queued: then dy must stay as it is until the flush, and later gradients of `res` go to a buffer of their own)
                self._acc(res, dy, dy_owned and not self._wg_on)
            if need_dx:
                wt = self.wt[w]
                Cin = x.shape[3]
                cur = self._slot(x) if mode != 1 else None
                if stride == 2:
                    dx = T.conv(dy, wt, Cin, taps, 1, 2, out=cur, accumulate=cur is not None)
                elif mode == 1:
                    dx = T.sum2x2(T.conv(dy, wt, Cin, taps, 1, 0))
                else:
                    dx = T.conv(dy, wt, Cin, taps, 1, 0, out=cur, accumulate=cur is not None)
                if cur is None:
                    self._acc(x, dx, True)
        self._tape.append(bwd)
        return y

    def _linear(self, x2d, name, need_dx=True, done=None):
        """x2d (B <= 16, K) -> (B, N): the row-wise kernels (one wave per output feature); more rows: a 1x1 conv over B
        one-pixel images"""
        w = name + ".weight"
        N = self.layers[w][0]
        done = done or (w, name + ".bias")
        B, K = x2d.shape
        if B > 16 or K % 8:
            return self._conv_view(x2d.view(B, 1, 1, K), x2d, name, need_dx, done)
        y = T.linear_rows(x2d, self.wf[w], N, bias=self.p[name + ".bias"])

        def bwd():
            self._wg_flush()                             # (dy may be the time-embedding-row sums of queued weight gradients)
            dy = self._pop(y)
            T.linear_rows_wgrad(dy, x2d, self.g[w], self.g[name + ".bias"])
            self._done(*done)
            if need_dx:
                self._acc(x2d, T.linear_rows(dy, self.wt[w], K), True)
        self._tape.append(bwd)
        return y

    def _conv_view(self, x4, x2d, name, need_dx, done):
        w = name + ".weight"
        N = self.layers[w][0]
        y4 = T.conv(x4, self.wf[w], N, 1, bias=self.p[name + ".bias"])
        y2 = y4.view(x2d.shape[0], N)

        def bwd():
            self._wg_flush()
            dy = self._pop(y2)
            dy4 = dy.view(dy.shape[0], 1, 1, N)
            T.wgrad(dy4, x4, self.g[w], 1)
            T.colsum(dy4, total=self.g[name + ".bias"])
            self._done(*done)
            if need_dx:
                dx = T.conv(dy4, self.wt[w], x2d.shape[1], 1)
                self._acc(x2d, dx.view(x2d.shape), True)
        self._tape.append(bwd)
        return y2

    def _gn(self, x, name, silu):
        cfg = self.cfg
        gamma, beta = self.p[name + ".weight"], self.p[name + ".bias"]
        y, stats = T.gn_forward(x, gamma, beta, cfg.norm_num_groups, cfg.norm_eps, silu)

        def bwd():
            dy = self._pop(y)
            cur = self._slot(x)
            dx = T.gn_backward(x, dy, stats, gamma, beta, cfg.norm_num_groups, silu, self.g[name + ".weight"], self.g[name + ".bias"],
                               dx=cur, accumulate=cur is not None)
            self._done(name + ".weight", name + ".bias")
            if cur is None:
                self._acc(x, dx, True)
        self._tape.append(bwd)
        return y

    def _silu(self, x):
        y = T.silu(x)

        def bwd():
            dy = self._pop(y)
            self._acc(x, T.silu_backward(x, dy), True)
        self._tape.append(bwd)
        return y

    def _concat(self, a, b):
        y = T.concat(a, b)
        ca, cb = a.shape[3], b.shape[3]

        def bwd():
            dy = self._pop(y)
            da, db = T.empty(a.shape, a), T.empty(b.shape, b)
            T.copy_channels(dy, 0, da, 0, ca)
            T.copy_channels(dy, ca, db, 0, cb)
            self._acc(a, da, True)
            self._acc(b, db, True)
        self._tape.append(bwd)
        return y

    def _resnet(self, x, p, temb_act):
        h = self._gn(x, p + ".norm1", True)
        row = self._rows[p + ".time_emb_proj"] if self._rows else self._linear(temb_act, p + ".time_emb_proj")
        h = self._conv(h, p + ".conv1", rowadd=row)
        h = self._gn(h, p + ".norm2", True)
        sc = self._conv(x, p + ".conv_shortcut") if (p + ".conv_shortcut.weight") in self.shapes else x
        return self._conv(h, p + ".conv2", res=sc)

    def _attention(self, x, p):
        B, W, H, Cc = x.shape
        y = self._gn(x, p + ".group_norm", False)
        if (p + ".to_qkv") in self.fused:
            grp = self.fused[p + ".to_qkv"]
            qkv = self._conv(y, p + ".to_qkv", done=grp["weights"] + grp["biases"])         # (B, W, H, 3C) = [q | k | v]
            qkv3 = qkv.view(B, W * H, 3 * Cc)
            o3, lse = T.attention_qkv_forward(qkv3)
            o = o3.view(B, W, H, Cc)

            def bwd():
                dO = self._pop(o)
                self._acc(qkv, T.attention_qkv_backward(qkv3, o3, dO.view(B, W * H, Cc), lse).view(qkv.shape), True)
            self._tape.append(bwd)
            return self._conv(o, p + ".to_out.0", res=x)
        q = self._conv(y, p + ".to_q")
        k = self._conv(y, p + ".to_k")
        v = self._conv(y, p + ".to_v")
        q3, k3, v3 = (t.view(B, W * H, Cc) for t in (q, k, v))
        o3, lse = T.attention_forward(q3, k3, v3)
        o = o3.view(B, W, H, Cc)

        def bwd():
            dO = self._pop(o)
            dq, dk, dv = T.attention_backward(q3, k3, v3, o3, dO.view(B, W * H, Cc), lse)
            self._acc(q, dq.view(q.shape), True)
            self._acc(k, dk.view(k.shape), True)
            self._acc(v, dv.view(v.shape), True)
        self._tape.append(bwd)
        return self._conv(o, p + ".to_out.0", res=x)

    # ---- fused tape (round 5) ---------------------------------------------------------------------------------
    class Cat:
        """`torch.cat([a, b], dim=1)` of the up blocks, never materialised: the consumers read the two sources in place."""
        __slots__ = ("a", "b")

        def __init__(self, a, b):
            self.a, self.b = a, b

    def fused_shape_ok(self, B, W, H):
        """Do the fused kernels cover every layer of this network at this input size?  (RLDM_TR_WG_V1, the A/B switch that makes the
        library refuse the all-taps weight-gradient kernel, also switches the fused blocks off: they have no other weight gradient)"""
        if os.environ.get("RLDM_TR_WG_V1"):
            return False
        return self.fused_tape and fused_tape_supported(self.cfg, self.fused, B, W, H)

    def _srcs(self, x):
        """The one or two source tensors of a (possibly concatenated) activation, each with its (sum, sumsq) pairs."""
        ts = (x.a, x.b) if isinstance(x, UNetTrainer.Cat) else (x,)
        out = []
        for t in ts:
            cs = self._cs_get(t)
            if cs is None:                              # no fused conv produced it (conv_in): one statistics launch
                cs = T.chan_stats(t)
                self._cs_set(t, cs)
            out.append(T.Src(t, cs))
        return out

    def _cs_set(self, t, cs):
        """The (sum, sumsq) pairs of tensor t, keyed by id(t) WITH a reference to t: the id cannot be recycled while the entry lives."""
        self._cs[id(t)] = (t, cs)

    def _cs_get(self, t):
        e = self._cs.get(id(t))
        return e[1] if e is not None and e[0] is t else None

    def _drow(self, rowadd):
        """Where the time-embedding row gradient of a conv goes: (buffer, accumulate?)"""
        parent = self._row_parent.get(id(rowadd))
        if parent is None:
            return T.empty(rowadd.shape, rowadd), False
        rows, off = parent
        if id(rows) not in self._grad:                   # zeroed once for all 22 slices (they accumulate into it)
            self._grad[id(rows)] = (torch.zeros(rows.shape, dtype=torch.float32, device=rows.device), True)
        return self._grad[id(rows)][0][:, off:off + rowadd.shape[1]], True

    def _resnet_f(self, x, p):
        """ResnetBlock2D in three launches: conv1 and conv2 normalise + activate their inputs while staging and accumulate their
        outputs' statistics; the 1x1 shortcut reads the (concatenated) input in place.  Backward: per conv one weight-gradient
        launch (operand rebuilt from the raw input) and one data-gradient launch (GroupNorm-backward transform + sums in the
        epilogue), per GroupNorm one apply launch (+ the residual-path gradient, split over the concatenation's sources)."""
        cfg = self.cfg
        srcs = self._srcs(x)
        Cin = sum(s.t.shape[3] for s in srcs)
        n1, n2, c1, c2, sc_n = p + ".norm1", p + ".norm2", p + ".conv1", p + ".conv2", p + ".conv_shortcut"
        N = self.layers[c1 + ".weight"][0]
        gn1 = T.GN(self.p[n1 + ".weight"], self.p[n1 + ".bias"], True, cfg.norm_num_groups, cfg.norm_eps)
        gn2 = T.GN(self.p[n2 + ".weight"], self.p[n2 + ".bias"], True, cfg.norm_num_groups, cfg.norm_eps)
        row = self._rows[p + ".time_emb_proj"]
        h1, cs1 = T.conv_fused(srcs, self.wf[c1 + ".weight"], N, 9, gn=gn1, bias=self.p[c1 + ".bias"], rowadd=row, want_stats=True)
        s1 = [T.Src(h1, cs1)]
        has_sc = (sc_n + ".weight") in self.shapes
        sc = T.conv_fused(srcs, self.wf[sc_n + ".weight"], N, 1, bias=self.p[sc_n + ".bias"]) if has_sc else srcs[0].t
        y, csy = T.conv_fused(s1, self.wf[c2 + ".weight"], N, 9, gn=gn2, bias=self.p[c2 + ".bias"], res=sc, want_stats=True)
        self._cs_set(y, csy)

        def bwd():
            dy = self._pop(y)
            T.wgrad_fused(dy, s1, self.g[c2 + ".weight"], 9, gn=gn2, total=self.g[c2 + ".bias"])
            self._pin(dy, h1, cs1)
            self._done(c2 + ".weight", c2 + ".bias")
            dz2, gs2 = T.conv_fused([T.Src(dy)], self.wt[c2 + ".weight"], N, 9, gsrcs=s1, ggn=gn2)
            dh1, = T.gn_backward_apply(dz2, s1, gs2, gn2, self.g[n2 + ".weight"], self.g[n2 + ".bias"])
            self._done(n2 + ".weight", n2 + ".bias")
            drow, racc = self._drow(row)
            T.wgrad_fused(dh1, srcs, self.g[c1 + ".weight"], 9, gn=gn1, rows=drow, total=self.g[c1 + ".bias"], rows_accumulate=racc)
            self._pin(dh1, drow, *[s.t for s in srcs], *[s.cs for s in srcs])
            self._done(c1 + ".weight", c1 + ".bias")
            if not racc:
                self._acc(row, drow, True)
            dz1, gs1 = T.conv_fused([T.Src(dh1)], self.wt[c1 + ".weight"], Cin, 9, gsrcs=srcs, ggn=gn1)
            if has_sc:
                T.wgrad_fused(dy, srcs, self.g[sc_n + ".weight"], 1, total=self.g[sc_n + ".bias"])
                self._pin(dy, *[s.t for s in srcs])
                self._done(sc_n + ".weight", sc_n + ".bias")
                res = T.conv(dy, self.wt[sc_n + ".weight"], Cin, 1)
            else:
                res = dy
            curs = [self._slot(s.t) for s in srcs]
            outs = T.gn_backward_apply(dz1, srcs, gs1, gn1, self.g[n1 + ".weight"], self.g[n1 + ".bias"], res=res, dsts=curs,
                                       accumulate=[c is not None for c in curs] + [False])
            self._done(n1 + ".weight", n1 + ".bias")
            for s, o, c in zip(srcs, outs, curs):
                if c is None:
                    self._acc(s.t, o, True)
        self._tape.append(bwd)
        return y

    def _attention_f(self, x, p):
        """Attention block: group_norm folded into the fused q/k/v projection's staging, to_out carries the residual and the
        output statistics; backward as _resnet_f (the projection's data gradient ends in the GroupNorm-backward epilogue)."""
        cfg = self.cfg
        B, W, H, Cc = x.shape
        srcs = self._srcs(x)
        gnn, wq, wo = p + ".group_norm", p + ".to_qkv", p + ".to_out.0"
        grp = self.fused[wq]
        gna = T.GN(self.p[gnn + ".weight"], self.p[gnn + ".bias"], False, cfg.norm_num_groups, cfg.norm_eps)
        qkv = T.conv_fused(srcs, self.wf[wq + ".weight"], 3 * Cc, 1, gn=gna, bias=self.p[wq + ".bias"])
        qkv3 = qkv.view(B, W * H, 3 * Cc)
        o3, lse = T.attention_qkv_forward(qkv3)
        o = o3.view(B, W, H, Cc)
        y, csy = T.conv_fused([T.Src(o)], self.wf[wo + ".weight"], Cc, 1, bias=self.p[wo + ".bias"], res=x, want_stats=True)
        self._cs_set(y, csy)

        def bwd():
            dy = self._pop(y)
            T.wgrad_bias(dy, o, self.g[wo + ".weight"], 1, total=self.g[wo + ".bias"])
            self._pin(dy, o)
            self._done(wo + ".weight", wo + ".bias")
            do = T.conv(dy, self.wt[wo + ".weight"], Cc, 1)
            dqkv = T.attention_qkv_backward(qkv3, o3, do.view(B, W * H, Cc), lse).view(qkv.shape)
            T.wgrad_fused(dqkv, srcs, self.g[wq + ".weight"], 1, gn=gna, total=self.g[wq + ".bias"])
            self._pin(dqkv, *[s.t for s in srcs], *[s.cs for s in srcs])
            self._done(*(grp["weights"] + grp["biases"]))
            dz, gs = T.conv_fused([T.Src(dqkv)], self.wt[wq + ".weight"], Cc, 1, gsrcs=srcs, ggn=gna)
            cur = self._slot(x)
            out, = T.gn_backward_apply(dz, srcs, gs, gna, self.g[gnn + ".weight"], self.g[gnn + ".bias"], res=dy, dsts=[cur],
                                       accumulate=[cur is not None, False])
            self._done(gnn + ".weight", gnn + ".bias")
            if cur is None:
                self._acc(x, out, True)
        self._tape.append(bwd)
        return y

    def _conv_stats(self, x, name, stride=1, mode=0):
        """A conv whose input is used as it is (down / up samplers) and whose output a GroupNorm will read: `_conv` with the
        output's statistics accumulated by the forward launch."""
        return self._conv(x, name, stride=stride, mode=mode, stats=True)

    # ---- network ----------------------------------------------------------------------------------------------
    def forward(self, sample_nchw, timesteps, pos_encoding=False):
        """sample (B, C, W, H) fp32 on the device; pos_encoding=True appends the azimuth-0 marker channel here (the
        `torch.cat([noisy_images, pos_encoding], dim=1)` of ldm/train_unconditional.py:500-501), so C + 1 == in_channels.
        timesteps (B,) int64 -> model_output (B, W, H, out_channels) NHWC.  Records the tape for `backward`."""
        cfg = self.cfg
        self._tape, self._grad, self._cs = [], {}, {}
        T.set_zero_arena(self._arena)                   # split-K conv outputs of this step: pre-zeroed slices, one fill
        self._arena.begin()
        fz = self.fused_shape_ok(sample_nchw.shape[0], sample_nchw.shape[2], sample_nchw.shape[3])
        self.last_forward_fused = fz
        x = T.pack_input(sample_nchw.float().contiguous(), pos_encoding)
        if x.shape[3] != cfg.in_channels:
            raise ValueError(f"sample has {x.shape[3]} channels (incl. pos-encoding), the UNet expects {cfg.in_channels}")
        B = x.shape[0]
        ts = timesteps.to(self.device, torch.int64).reshape(-1)
        if ts.numel() == 1:
            ts = ts.expand(B).contiguous()
        e = T.timestep_embedding(ts, cfg.block_out_channels[0])
        e = self._linear(e, "time_embedding.linear_1", need_dx=False)
        temb = self._linear(self._silu(e), "time_embedding.linear_2")
        temb_act = self._silu(temb)
        self._rows, self._row_parent = {}, {}
        if "time_emb_proj_all" in self.fused:           # every resnet's time_emb_proj in one launch; the resnets add column slices
            grp = self.fused["time_emb_proj_all"]
            rows = self._linear(temb_act, "time_emb_proj_all", done=grp["weights"] + grp["biases"])
            off = 0
            for wn in grp["weights"]:
                view = rows[:, off:off + self.shapes[wn][0]]
                self._rows[wn[:-len(".weight")]] = view
                self._row_parent[id(view)] = (rows, off)
                off += self.shapes[wn][0]
        h = self._conv(x, "conv_in", need_dx=False)
        skips = [h]
        nl = len(cfg.block_out_channels)
        # per level: the fused blocks where they pay (measured per shape, profiles/round5_train_by_grid_*: at >= `fused_min_pixels`
        # pixels per image the GroupNorm passes they remove cost more than what they add to the conv launches; below, a launch of
        # either kind sits at its ~8 us floor and the op-per-layer blocks are as fast)
        W0, H0 = sample_nchw.shape[2], sample_nchw.shape[3]
        fzl = [fz and on for on in fused_levels(W0, H0, nl, self.fused_min_pixels)]
        self.last_forward_fused_levels = fzl

        def resnet(t, p, l):
            if fzl[l]:
                return self._resnet_f(t, p)
            if isinstance(t, UNetTrainer.Cat):
                t = self._concat(t.a, t.b)
            return self._resnet(t, p, temb_act)

        def attention(t, p, l):
            return self._attention_f(t, p) if fzl[l] else self._attention(t, p)

        for i, bt in enumerate(cfg.down_block_types):
            for j in range(cfg.layers_per_block):
                h = resnet(h, f"down_blocks.{i}.resnets.{j}", i)
                if bt == "AttnDownBlock2D":
                    h = attention(h, f"down_blocks.{i}.attentions.{j}", i)
                skips.append(h)
            if i != nl - 1:
                # (the output's statistics ride in the conv's epilogue when the next level's blocks will ask for them)
                h = self._conv(h, f"down_blocks.{i}.downsamplers.0.conv", stride=2, stats=fzl[i + 1])
                skips.append(h)
        h = resnet(h, "mid_block.resnets.0", nl - 1)
        if cfg.add_attention:
            h = attention(h, "mid_block.attentions.0", nl - 1)
        h = resnet(h, "mid_block.resnets.1", nl - 1)
        for i, bt in enumerate(cfg.up_block_types):
            l = nl - 1 - i
            for j in range(cfg.layers_per_block + 1):
                h = UNetTrainer.Cat(h, skips.pop())
                h = resnet(h, f"up_blocks.{i}.resnets.{j}", l)
                if bt == "AttnUpBlock2D":
                    h = attention(h, f"up_blocks.{i}.attentions.{j}", l)
            if i != nl - 1:
                h = self._conv(h, f"up_blocks.{i}.upsamplers.0.conv", mode=1, stats=fzl[l - 1])
        assert not skips
        h = self._gn(h, "conv_norm_out", True)
        self._out = self._conv(h, "conv_out")
        return self._out

    def backward(self, dpred, reduce=None, launch_collectives=True):
        """Replays the tape in reverse.  reduce: None = single process; True = bucketed RCCL all-reduce (average) of the flat
        gradient buffer, overlapped with the rest of backward."""
        if reduce is None:
            reduce = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        self._pending = []
        self._ready = None
        avg = False
        if reduce:
            self._pidx = {n: i for i, n in enumerate(self.names)}
            self._ready = [hi - lo for lo, hi, _, _ in self.buckets]
            from . import distributed as D
            # RCCL (torch's "nccl" backend, or the C-ABI communicator) averages in the collective; gloo sums
            avg = torch.distributed.get_backend() == "nccl" or D.cabi_communicator() is not None
            self._reduce_op = torch.distributed.ReduceOp.AVG if avg else torch.distributed.ReduceOp.SUM
        self._acc(self._out, dpred, False)
        T.defer_reduce(self.defer_reduce)
        self._wg_on = bool(self.wgrad_group)
        T.wgrad_group(self._wg_on)
        try:
            for fn in reversed(self._tape):
                fn()
        finally:
            T.wgrad_group(False)                        # (runs what is still queued)
            self._wg_on = False
            self._wg_keep.clear()
            T.defer_reduce(False)                       # (flushes: every gradient is final from here on)
        self._tape, self._grad, self._cs = [], {}, {}
        T.set_zero_arena(None)
        if launch_collectives:
            for w in self._pending:
                w.wait()
        self._ready = None
        return torch.distributed.get_world_size() if (reduce and not avg) else 1

    def optimizer_step(self, grad_world=1, device_scalars=False):
        """clip_grad_norm_(1.0) + AdamW + cosine/warmup lr + EMA (ldm/train_unconditional.py:546-556), then refresh the
        bf16 operand copies and zero the gradients.  device_scalars: the step number lives in device memory and the
        learning rate / bias corrections / EMA decay are computed there (what a captured step graph replays)."""
        hp = self.hp
        if device_scalars and self._step_dev_mirror != self.global_step:
            self._step_dev.fill_(self.global_step)
            self._step_dev_mirror = self.global_step
        self.global_step += 1
        if grad_world > 1:                              # a summing backend (gloo): finish the average
            self.grads.mul_(1.0 / grad_world)
        sq = T.sqnorm(self.grads)
        if device_scalars:
            self._step_dev_mirror += 1
            T.hyper_step(self._step_dev, self._dyn, hp["lr"], hp["betas"], hp["ema_max_decay"], hp["ema_inv_gamma"],
                         hp["ema_power"], hp["lr_warmup_steps"], hp["total_steps"])
            T.adamw_dyn(self.params, self.grads, self.exp_avg, self.exp_avg_sq, self._dyn, hp["betas"], hp["eps"],
                        hp["weight_decay"], ema=self.ema, sqnorm_dev=sq, max_grad_norm=hp["max_grad_norm"], zero_grads=True)
            self.last_grad_norm = sq
            self.repack()
            return None
        lr = cosine_lr(self.global_step - 1, hp["lr"], hp["lr_warmup_steps"], hp["total_steps"])
        # (lr_scheduler.step() runs AFTER optimizer.step(): step k uses the rate of k - 1 scheduler steps)
        dec = ema_decay(self.global_step, hp["ema_max_decay"], hp["ema_inv_gamma"], hp["ema_power"])
        T.adamw(self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.global_step, lr, hp["betas"], hp["eps"],
                hp["weight_decay"], ema=self.ema, ema_decay=dec, sqnorm_dev=sq, max_grad_norm=hp["max_grad_norm"])
        self.last_grad_norm = sq
        self.repack()
        self.grads.zero_()
        return lr

    def train_step(self, noisy_nchw, timesteps, target_nchw, loss_weights=None, pos_encoding=False):
        """model_output = model(noisy, t); loss = mse(model_output, target) [* min-SNR weights]; backward; step.
        Returns the loss (0-d float64 device tensor; no host sync)."""
        pred = self.forward(noisy_nchw, timesteps, pos_encoding)
        loss, dpred = T.mse(pred, target_nchw.float().contiguous(), loss_weights)
        k = self.gradient_accumulation_steps
        self._micro += 1
        if self._micro < k:
            # a micro-batch inside the accumulation window: gradients add up in the flat buffer, no exchange with the other
            # ranks (accelerate's no_sync) and no optimizer step
            self.backward(dpred, reduce=False)
            return loss
        self._micro = 0
        world = self.backward(dpred)
        self.optimizer_step(world * k)                   # mean over ranks and over the k micro-batches
        return loss

    # ---- captured step graphs ---------------------------------------------------------------------------------
    def train_step_graphed(self, noisy_nchw, timesteps, target_nchw, loss_weights=None, pos_encoding=False, reduce=None):
        """`train_step` replayed from HIP graphs: the ~1 300 launches of a step cost more host time to enqueue than the GPU
        needs to run them.  The first call of a shape runs eagerly (it sizes the library's scratch buffers), the second
        captures, later ones copy their inputs into the graph's static buffers and replay.  With more than one rank the
        step is cut into one graph per gradient bucket: after each, the bucket's RCCL all-reduce is launched eagerly on
        the communication stream and overlaps the next segment of backward; the optimizer segment waits for all of them.
        Returns the loss (0-d float64 device tensor, a copy)."""
        if self.gradient_accumulation_steps != 1:
            raise NotImplementedError("train_step_graphed captures one optimizer step per call; use train_step with "
                                      "gradient_accumulation_steps > 1")
        dist = torch.distributed
        if reduce is None:
            reduce = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        key = (tuple(noisy_nchw.shape), tuple(target_nchw.shape), loss_weights is not None, bool(pos_encoding), bool(reduce))
        st = self._graphs.get(key)
        if st is None:
            self._graphs[key] = "warm"
            return self.train_step(noisy_nchw, timesteps, target_nchw, loss_weights, pos_encoding)
        if st == "warm":
            st = self._graphs[key] = self._capture_step(noisy_nchw, timesteps, target_nchw, loss_weights, pos_encoding, reduce)
        st["noisy"].copy_(noisy_nchw, non_blocking=True)
        st["t"].copy_(timesteps.reshape(-1), non_blocking=True)
        st["target"].copy_(target_nchw, non_blocking=True)
        if loss_weights is not None:
            st["w"].copy_(loss_weights, non_blocking=True)
        if self._step_dev_mirror != self.global_step:
            self._step_dev.fill_(self.global_step)
            self._step_dev_mirror = self.global_step
        pending = []
        for graph, action in st["segments"]:
            graph.replay()
            if action == "wait":
                for w in pending:
                    w.wait()
            elif action is not None:
                _, _, off, cnt = self.buckets[action]
                pending.append(self._launch_reduce(off, cnt, st["reduce_op"]))
        self.global_step += 1
        self._step_dev_mirror += 1
        self.last_grad_norm = st["sqnorm"]
        return st["loss"].clone()

    def _capture_step(self, noisy, timesteps, target, loss_weights, pos_encoding, reduce):
        st = {"noisy": noisy.detach().float().clone(), "t": timesteps.to(self.device, torch.int64).reshape(-1).clone(),
              "target": target.detach().float().clone(), "w": None if loss_weights is None else loss_weights.detach().float().clone(),
              "segments": [], "reduce_op": None}
        pool = torch.cuda.graph_pool_handle()
        tick = torch.zeros(1, device=self.device)
        if self._step_dev_mirror != self.global_step:
            self._step_dev.fill_(self.global_step)
            self._step_dev_mirror = self.global_step
        step0, mirror0 = self.global_step, self._step_dev_mirror
        cap = torch.cuda.Stream(self.device)
        cap.wait_stream(torch.cuda.current_stream(self.device))
        cur = [None]

        def begin():
            cur[0] = torch.cuda.CUDAGraph()
            cur[0].capture_begin(pool=pool)
            tick.zero_()                                  # (a segment is never empty)

        def cut(action):
            cur[0].capture_end()
            st["segments"].append((cur[0], action))

        def on_bucket(b):
            cut(b)
            begin()

        with torch.cuda.stream(cap):
            begin()
            try:
                self._on_bucket = on_bucket if reduce else None
                pred = self.forward(st["noisy"], st["t"], pos_encoding)
                loss, dpred = T.mse(pred, st["target"], st["w"])
                world = self.backward(dpred, reduce=reduce, launch_collectives=False)
                if reduce:
                    on_bucket("wait")
                self.optimizer_step(world, device_scalars=True)
                cut(None)
            finally:
                self._on_bucket = None
        torch.cuda.current_stream(self.device).wait_stream(cap)
        # capture launched nothing: the step counters still describe the state before this step
        self.global_step, self._step_dev_mirror = step0, mirror0
        st["loss"], st["sqnorm"], st["tick"], st["pool"] = loss, self.last_grad_norm, tick, pool
        if reduce:
            st["reduce_op"] = self._reduce_op
        return st


def encode_ahead(vae, clean_images, stream, generator=None):
    """The VAE half of the loop body (ldm/train_unconditional.py:480-481) for the NEXT batch, enqueued on `stream` so that it
    runs beside the current step's many small UNet launches.  Returns (latents, event): wait for the event on the training
    stream, then pass `latents=` to training_step."""
    cur = torch.cuda.current_stream(clean_images.device if clean_images.is_cuda else None)
    stream.wait_stream(cur)                                    # the images were produced on the caller's stream
    with torch.cuda.stream(stream):
        latents = vae.encode(clean_images).latent_dist.sample(generator=generator, scale=vae.config.scaling_factor)
        ev = torch.cuda.Event()
        ev.record(stream)
    latents.record_stream(cur)
    return latents, ev


def training_step(trainer, vae, noise_scheduler, clean_images, generator=None, pos_encoding=True, snr_gamma=None,
                  noise=None, timesteps=None, condition=None, graphed=False, latents=None):
    """One iteration of the reference's loop body (ldm/train_unconditional.py:479-556) with `with_vae: True`:
    latents = vae.encode(x).latent_dist.sample() * scaling_factor; eps ~ N(0, 1); t ~ U{0..T-1}; add_noise; pos-encoding
    channel; epsilon- or v-prediction MSE (optionally min-SNR weighted); backward; clip; AdamW; lr schedule; EMA.
    condition (B, Cc, W, H): the conditional twin (ldm/train_conditional.py:418-447) -- the encoded low-resolution image
    (`condition_encoder(batch["down"])`) or `cat([masked latents, mask])`, concatenated to the noisy latents.
    graphed: replay the UNet forward / backward / optimizer from captured HIP graphs (UNetTrainer.train_step_graphed).
    latents: `vae.encode(clean_images).latent_dist.sample() * scaling_factor` computed by the caller (input pipelining:
    `encode_ahead`); clean_images is ignored then."""
    dev = trainer.device
    if latents is not None:                 # encoded ahead of time (e.g. on a second stream while the previous step ran)
        latents = latents.to(dev).float()
    elif vae is not None:
        latents = vae.encode(clean_images.to(dev)).latent_dist.sample(generator=generator, scale=vae.config.scaling_factor)
    else:
        latents = clean_images.to(dev).float()
    B = latents.shape[0]
    if noise is None:
        noise = torch.randn(latents.shape, generator=generator, dtype=torch.float32).to(dev)
    if timesteps is None:
        timesteps = torch.randint(0, noise_scheduler.config.num_train_timesteps, (B,), generator=generator).long()
    noisy = noise_scheduler.add_noise(latents, noise, timesteps)
    # the regression target (ldm/train_unconditional.py:505-510): the noise, or the velocity for a v_prediction scheduler
    ptype = getattr(noise_scheduler.config, "prediction_type", "epsilon")
    if ptype == "epsilon":
        target = noise
    elif ptype == "v_prediction":
        target = noise_scheduler.get_velocity(latents, noise, timesteps)
    else:
        raise ValueError(f"Unknown prediction type {ptype}")
    if condition is not None:
        noisy = torch.cat([noisy, condition.to(dev).float()], dim=1)        # (a copy: ldm/train_conditional.py:447)
    w = None
    if snr_gamma is not None:
        w = snr_weights(noise_scheduler.alphas_cumprod, timesteps, snr_gamma, v_prediction=ptype == "v_prediction").to(dev)
    step = trainer.train_step_graphed if graphed else trainer.train_step
    return step(noisy, timesteps.to(dev), target, w, pos_encoding)
