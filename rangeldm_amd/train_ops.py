"""Thin wrappers over the op-level training entry points of librangeldm_hip (rangeldm_amd/csrc/train.hip).

Tensors are device fp32, activations / gradients channels-last (B, W, H, C).  torch is used for allocation and streams
only; every arithmetic operation below is a HIP kernel behind the C ABI.  No CPU fallback.
"""
import ctypes as C

import torch

from . import _lib


def _s(t):
    return _lib.stream_ptr(t.device)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(rc, what):
    _lib.check(rc, what)


def empty(shape, like=None, device=None):
    return torch.empty(shape, dtype=torch.float32, device=like.device if like is not None else device)


def conv_desc(x, N, taps, stride=1, mode=0):
    d = _lib.TrainConvDescC()
    d.B, d.Win, d.Hin, d.Cin = x.shape
    d.N, d.taps, d.stride, d.mode = N, taps, stride, mode
    return d


def out_size(W, H, stride, mode):
    sh = 1 if mode else 0
    return (W << sh) // stride, (H << sh) // stride


def pack_weights(w, taps, want_transposed=True):
    """master fp32 (N, Cin, k, k) or (N, Cin) -> (bf16 [N][taps][ceil16 Cin], bf16 [Cin][taps][ceil16 N] flipped)"""
    N, Cin = w.shape[0], w.shape[1]
    cp, npad = (Cin + 15) // 16 * 16, (N + 15) // 16 * 16
    wf = torch.empty((N, taps, cp), dtype=torch.bfloat16, device=w.device)
    wt = torch.empty((Cin, taps, npad), dtype=torch.bfloat16, device=w.device) if want_transposed else None
    _chk(_lib.lib().rldm_train_pack_weights(_p(w), N, Cin, taps, _p(wf), _p(wt), _s(w)), "rldm_train_pack_weights")
    return wf, wt


class ZeroArena:
    """Pre-zeroed outputs for the split-K conv launches of one step: every such launch gets its own slice (handed out in call
    order, the same every step), and ONE fill at the start of the step replaces a fill launch per conv (75 per step at the
    RangeLDM size).  `begin()` zeroes what the previous step used and rewinds; `take(shape)` returns the next slice."""

    def __init__(self, device, nbytes=64 << 20):
        self.device, self.buf, self.pos, self.high = device, None, 0, 0
        self.cap = nbytes // 4

    def begin(self):
        if self.buf is None:
            self.buf = torch.zeros(self.cap, dtype=torch.float32, device=self.device)
        elif self.high:
            self.buf[:self.high].zero_()
        self.pos = 0

    def take(self, shape):
        n = 1
        for v in shape:
            n *= int(v)
        n4 = (n + 3) // 4 * 4
        if self.buf is None or self.pos + n4 > self.cap:
            return None                                  # arena full: the launch zero-fills its own output
        t = self.buf[self.pos:self.pos + n].view(shape)
        self.pos += n4
        self.high = max(self.high, self.pos)
        return t


_arena = None


def set_zero_arena(arena):
    """Install (or remove, None) the arena `conv` draws split-K outputs from."""
    global _arena
    _arena = arena


def conv(x, w_packed, N, taps, stride=1, mode=0, bias=None, rowadd=None, res=None, out=None, accumulate=False):
    """y = conv(x) + bias + rowadd[b] + res.  x (B, W, H, Cin) -> (B, Wo, Ho, N)."""
    d = conv_desc(x, N, taps, stride, mode)
    Wo, Ho = out_size(x.shape[1], x.shape[2], stride, mode)
    y = out
    if y is None and _arena is not None and \
            _lib.lib().rldm_train_conv_splits(C.byref(d), 0 if rowadd is None else rowadd.stride(0)) > 1:
        y = _arena.take((x.shape[0], Wo, Ho, N))        # pre-zeroed: the split launch only adds
        accumulate = y is not None
    if y is None:
        y = empty((x.shape[0], Wo, Ho, N), x)
    _chk(_lib.lib().rldm_train_conv(C.byref(d), _p(x), _p(w_packed), _p(bias), _p(rowadd),
                                    0 if rowadd is None else rowadd.stride(0), _p(res), _p(y), 1 if accumulate else 0, _s(x)),
         "rldm_train_conv")
    return y


def wgrad(dy, x, dw, taps, stride=1, mode=0):
    """dw (N, Cin, k, k) += dy (x) x  (dw is a view into the zeroed flat gradient buffer)."""
    d = conv_desc(x, dy.shape[3], taps, stride, mode)
    _chk(_lib.lib().rldm_train_wgrad(C.byref(d), _p(dy), _p(x), _p(dw), _s(x)), "rldm_train_wgrad")


def wgrad_bias(dy, x, dw, taps, stride=1, mode=0, rows=None, total=None, rows_accumulate=False):
    """wgrad + colsum in one call (one launch where the all-taps kernel applies)."""
    d = conv_desc(x, dy.shape[3], taps, stride, mode)
    _chk(_lib.lib().rldm_train_wgrad_bias(C.byref(d), _p(dy), _p(x), _p(dw), _p(rows), 0 if rows is None else rows.stride(0),
                                          1 if rows_accumulate else 0, _p(total), _s(x)), "rldm_train_wgrad_bias")


def colsum(dy, rows=None, total=None, rows_accumulate=False):
    B, W, H, N = dy.shape
    _chk(_lib.lib().rldm_train_colsum(_p(dy), B, W * H, N, _p(rows), 0 if rows is None else rows.stride(0),
                                      1 if rows_accumulate else 0, _p(total), _s(dy)), "rldm_train_colsum")


def gn_forward(x, gamma, beta, groups, eps, silu):
    B, W, H, Cc = x.shape
    stats = empty((B, groups, 2), x)
    y = torch.empty_like(x)
    _chk(_lib.lib().rldm_train_gn_forward(_p(x), B, W * H, Cc, groups, float(eps), _p(gamma), _p(beta), 1 if silu else 0,
                                          _p(stats), _p(y), _s(x)), "rldm_train_gn_forward")
    return y, stats


def gn_backward(x, dy, stats, gamma, beta, groups, silu, dgamma, dbeta, dx=None, accumulate=False):
    B, W, H, Cc = x.shape
    scratch = empty((B, groups, 2), x)
    dx = dx if dx is not None else torch.empty_like(x)
    _chk(_lib.lib().rldm_train_gn_backward(_p(x), _p(dy), _p(stats), B, W * H, Cc, groups, _p(gamma), _p(beta),
                                           1 if silu else 0, _p(scratch), _p(dx), 1 if accumulate else 0, _p(dgamma),
                                           _p(dbeta), _s(x)), "rldm_train_gn_backward")
    return dx


def attention_forward(q, k, v):
    """q, k, v (B, L, C) -> o (B, L, C), lse (B, C/8, L)"""
    B, L, Cc = q.shape
    o = torch.empty_like(q)
    lse = empty((B, Cc // 8, L), q)
    _chk(_lib.lib().rldm_train_attention_forward(_p(q), _p(k), _p(v), B, L, Cc, _p(o), _p(lse), _s(q)),
         "rldm_train_attention_forward")
    return o, lse


def attention_backward(q, k, v, o, dO, lse):
    B, L, Cc = q.shape
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    delta = torch.empty_like(lse)
    _chk(_lib.lib().rldm_train_attention_backward(_p(q), _p(k), _p(v), _p(o), _p(dO), _p(lse), B, L, Cc, _p(delta), _p(dq),
                                                  _p(dk), _p(dv), _s(q)), "rldm_train_attention_backward")
    return dq, dk, dv


def linear_rows(x, w_packed, N, bias=None, out=None, accumulate=False):
    """x (B <= 16, K) fp32 rows (may be a column slice of a wider matrix) -> (B, N) = x W^T + bias; W bf16 [N][ceil16 K]."""
    B, K = x.shape
    y = out if out is not None else empty((B, N), x)
    _chk(_lib.lib().rldm_train_linear_rows(_p(x), x.stride(0), _p(w_packed), K, _p(bias), _p(y), y.stride(0), B, N,
                                           1 if accumulate else 0, _s(x)), "rldm_train_linear_rows")
    return y


def linear_rows_wgrad(dy, x, dw, dbias=None):
    """dw (N, K) += dy^T x; dbias (N,) += dy.sum(0)"""
    B, N = dy.shape
    _chk(_lib.lib().rldm_train_linear_rows_wgrad(_p(dy), dy.stride(0), _p(x), x.stride(0), B, N, x.shape[1], _p(dw), _p(dbias),
                                                 _s(x)), "rldm_train_linear_rows_wgrad")


def attention_qkv_forward(qkv):
    """qkv (B, L, 3C) = [q | k | v] -> o (B, L, C), lse (B, C/8, L)"""
    B, L, C3 = qkv.shape
    Cc = C3 // 3
    o = empty((B, L, Cc), qkv)
    lse = empty((B, Cc // 8, L), qkv)
    _chk(_lib.lib().rldm_train_attention_qkv_forward(_p(qkv), B, L, Cc, _p(o), _p(lse), _s(qkv)), "rldm_train_attention_qkv_forward")
    return o, lse


def attention_qkv_backward(qkv, o, dO, lse):
    """-> dqkv (B, L, 3C) = [dq | dk | dv]"""
    B, L, C3 = qkv.shape
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    _chk(_lib.lib().rldm_train_attention_qkv_backward(_p(qkv), _p(o), _p(dO), _p(lse), B, L, C3 // 3, _p(delta), _p(dqkv), _s(qkv)),
         "rldm_train_attention_qkv_backward")
    return dqkv


def add(a, b, out=None):
    y = out if out is not None else torch.empty_like(a)
    _chk(_lib.lib().rldm_train_add(_p(a), _p(b), _p(y), a.numel(), _s(a)), "rldm_train_add")
    return y


def copy_channels(src, src_off, dst, dst_off, ncopy, accumulate=False):
    npix = src.numel() // src.shape[-1]
    _chk(_lib.lib().rldm_train_copy_channels(_p(src), src.shape[-1], src_off, _p(dst), dst.shape[-1], dst_off, ncopy, npix,
                                             1 if accumulate else 0, _s(src)), "rldm_train_copy_channels")


def concat(a, b):
    y = empty((*a.shape[:-1], a.shape[-1] + b.shape[-1]), a)
    copy_channels(a, 0, y, 0, a.shape[-1])
    copy_channels(b, 0, y, a.shape[-1], b.shape[-1])
    return y


def sum2x2(du):
    B, W2, H2, Cc = du.shape
    dx = empty((B, W2 // 2, H2 // 2, Cc), du)
    _chk(_lib.lib().rldm_train_sum2x2(_p(du), B, W2 // 2, H2 // 2, Cc, _p(dx), _s(du)), "rldm_train_sum2x2")
    return dx


def silu(x):
    y = torch.empty_like(x)
    _chk(_lib.lib().rldm_train_silu(_p(x), None, _p(y), x.numel(), 0, 0, _s(x)), "rldm_train_silu")
    return y


def silu_backward(x, dy, out=None, accumulate=False):
    y = out if out is not None else torch.empty_like(x)
    _chk(_lib.lib().rldm_train_silu(_p(x), _p(dy), _p(y), x.numel(), 1, 1 if accumulate else 0, _s(x)), "rldm_train_silu")
    return y


def timestep_embedding(timesteps, dim):
    t = timesteps.to(torch.int64).contiguous()
    out = empty((t.shape[0], dim), t)
    _chk(_lib.lib().rldm_train_timestep_embedding(_p(t), t.shape[0], dim, _p(out), _s(t)), "rldm_train_timestep_embedding")
    return out


def pack_input(x_nchw, pos_encoding):
    B, Cc, W, H = x_nchw.shape
    y = empty((B, W, H, Cc + (1 if pos_encoding else 0)), x_nchw)
    _chk(_lib.lib().rldm_train_pack_input(_p(x_nchw), B, Cc, W, H, 1 if pos_encoding else 0, _p(y), _s(x_nchw)),
         "rldm_train_pack_input")
    return y


def unpack_output(y_nhwc):
    B, W, H, Cc = y_nhwc.shape
    out = empty((B, Cc, W, H), y_nhwc)
    _chk(_lib.lib().rldm_train_unpack_output(_p(y_nhwc), B, Cc, W, H, _p(out), _s(y_nhwc)), "rldm_train_unpack_output")
    return out


def mse(pred_nhwc, target_nchw, weight=None):
    """-> (loss: 0-d float64 device tensor, dpred (B, W, H, C))"""
    B, W, H, Cc = pred_nhwc.shape
    dpred = torch.empty_like(pred_nhwc)
    loss = torch.empty((), dtype=torch.float64, device=pred_nhwc.device)
    _chk(_lib.lib().rldm_train_mse(_p(pred_nhwc), _p(target_nchw), _p(weight), B, Cc, W, H, _p(dpred), _p(loss),
                                   _s(pred_nhwc)), "rldm_train_mse")
    return loss, dpred


def sqnorm(g):
    out = torch.empty((), dtype=torch.float64, device=g.device)
    _chk(_lib.lib().rldm_train_sqnorm(_p(g), g.numel(), _p(out), _s(g)), "rldm_train_sqnorm")
    return out


def adamw(params, grads, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, ema=None,
          ema_decay=0.0, sqnorm_dev=None, max_grad_norm=0.0):
    c = _lib.AdamWConfigC()
    c.lr, c.beta1, c.beta2, c.eps, c.weight_decay = float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay)
    c.max_grad_norm, c.ema_decay, c.step = float(max_grad_norm), float(ema_decay), int(step)
    _chk(_lib.lib().rldm_train_adamw(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), _p(ema), _p(sqnorm_dev),
                                     params.numel(), C.byref(c), _s(params)), "rldm_train_adamw")


def hyper_step(step_counter, dyn, lr, betas, ema_max_decay, ema_inv_gamma, ema_power, lr_warmup_steps, total_steps):
    """step = ++step_counter (device int64); dyn (4,) <- (lr, 1 - b1^step, 1 - b2^step, ema decay) of that optimizer step."""
    c = _lib.HyperConfigC()
    c.lr, c.beta1, c.beta2 = float(lr), float(betas[0]), float(betas[1])
    c.ema_max_decay, c.ema_inv_gamma, c.ema_power = float(ema_max_decay), float(ema_inv_gamma), float(ema_power)
    c.lr_warmup_steps, c.total_steps = int(lr_warmup_steps), int(total_steps)
    _chk(_lib.lib().rldm_train_hyper_step(_p(step_counter), C.byref(c), _p(dyn), _s(dyn)), "rldm_train_hyper_step")


def adamw_dyn(params, grads, exp_avg, exp_avg_sq, dyn, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, ema=None,
              sqnorm_dev=None, max_grad_norm=0.0, zero_grads=True):
    """AdamW (+ clip + EMA) with lr / bias corrections / EMA decay read from the device vector `dyn` (see hyper_step)."""
    c = _lib.AdamWConfigC()
    c.lr, c.beta1, c.beta2, c.eps, c.weight_decay = 0.0, float(betas[0]), float(betas[1]), float(eps), float(weight_decay)
    c.max_grad_norm, c.ema_decay, c.step = float(max_grad_norm), 0.0, 1
    _chk(_lib.lib().rldm_train_adamw_dyn(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), _p(ema), _p(sqnorm_dev),
                                         params.numel(), C.byref(c), _p(dyn), 1 if zero_grads else 0, _s(params)),
         "rldm_train_adamw_dyn")


# ---- fused tape (round 5; include/rangeldm_hip.h "fused tape") ------------------------------------------------------------
class Src:
    """A forward tensor together with the per-(image, channel) (sum, sumsq) pairs its producer accumulated (None: not known)."""
    __slots__ = ("t", "cs")

    def __init__(self, t, cs=None):
        self.t, self.cs = t, cs


class GN:
    """A GroupNorm (+ SiLU) to be rebuilt by the consumers of its input: affine parameters + hyper-parameters."""
    __slots__ = ("gamma", "beta", "silu", "groups", "eps")

    def __init__(self, gamma, beta, silu, groups, eps):
        self.gamma, self.beta, self.silu, self.groups, self.eps = gamma, beta, bool(silu), int(groups), float(eps)


def zeros_stats(B, Cc, like):
    """A zeroed [B][C][2] accumulator: a slice of the step's zero arena when one is installed (one fill per step)."""
    t = _arena.take((B, Cc, 2)) if _arena is not None else None
    return t if t is not None else torch.zeros((B, Cc, 2), dtype=torch.float32, device=like.device)


def _fuse(srcs=None, gn=None, cs_out=None, gsrcs=None, ggn=None, gs_out=None):
    """-> (TrainFuseC, keep-alive list)"""
    f = _lib.TrainFuseC()
    keep = []
    if srcs is not None and len(srcs) == 2:
        f.x1, f.C0 = srcs[1].t.data_ptr(), srcs[0].t.shape[-1]
        keep.append(srcs[1].t)
    if gn is not None:
        f.cs0 = srcs[0].cs.data_ptr()
        f.cs1 = srcs[1].cs.data_ptr() if len(srcs) == 2 else None
        f.gamma, f.beta, f.silu, f.groups, f.eps = gn.gamma.data_ptr(), gn.beta.data_ptr(), 1 if gn.silu else 0, gn.groups, gn.eps
        keep += [s.cs for s in srcs] + [gn.gamma, gn.beta]
    if cs_out is not None:
        f.cs_out = cs_out.data_ptr()
        keep.append(cs_out)
    if gs_out is not None:
        f.g0, f.gcs0 = gsrcs[0].t.data_ptr(), gsrcs[0].cs.data_ptr()
        if len(gsrcs) == 2:
            f.g1, f.gcs1, f.G0 = gsrcs[1].t.data_ptr(), gsrcs[1].cs.data_ptr(), gsrcs[0].t.shape[-1]
        f.ggamma, f.gbeta, f.gsilu, f.ggroups, f.geps = (ggn.gamma.data_ptr(), ggn.beta.data_ptr(), 1 if ggn.silu else 0, ggn.groups,
                                                         ggn.eps)
        f.gs_out = gs_out.data_ptr()
        keep += [s.t for s in gsrcs] + [s.cs for s in gsrcs] + [ggn.gamma, ggn.beta, gs_out]
    return f, keep


def _cin(srcs):
    return sum(s.t.shape[-1] for s in srcs)


def _desc_srcs(srcs, N, taps, stride, mode):
    d = _lib.TrainConvDescC()
    d.B, d.Win, d.Hin = srcs[0].t.shape[:3]
    d.Cin = _cin(srcs)
    d.N, d.taps, d.stride, d.mode = N, taps, stride, mode
    return d


def conv_fused_ok(srcs, N, taps, stride=1, mode=0, gn=None, want_stats=False, gsrcs=None, ggn=None, rowadd=None):
    d = _desc_srcs(srcs, N, taps, stride, mode)
    dummy = srcs[0].t
    f, _ = _fuse(srcs, gn, dummy if want_stats else None, gsrcs, ggn, dummy if gsrcs is not None else None)
    return bool(_lib.lib().rldm_train_conv_fused_ok(C.byref(d), C.byref(f), 0 if rowadd is None else rowadd.stride(0)))


def conv_fused(srcs, w_packed, N, taps, stride=1, mode=0, gn=None, bias=None, rowadd=None, res=None, want_stats=False,
               gsrcs=None, ggn=None):
    """y = conv(act(GN(cat(srcs)))) + bias + rowadd[b] + res with everything folded into one launch.
    want_stats: -> (y, cs_y).  gsrcs / ggn (data gradient): y is stored as dz = y act'(z) -> (dz, gs)."""
    x = srcs[0].t
    B = x.shape[0]
    d = _desc_srcs(srcs, N, taps, stride, mode)
    Wo, Ho = out_size(x.shape[1], x.shape[2], stride, mode)
    out2 = zeros_stats(B, N, x) if (want_stats or gsrcs is not None) else None
    f, keep = _fuse(srcs, gn, out2 if want_stats else None, gsrcs, ggn, out2 if gsrcs is not None else None)
    y, pre = None, 0
    if _arena is not None and _lib.lib().rldm_train_conv_splits(C.byref(d), 0 if rowadd is None else rowadd.stride(0)) > 1:
        y = _arena.take((B, Wo, Ho, N))
        pre = 1 if y is not None else 0
    if y is None:
        y = empty((B, Wo, Ho, N), x)
    _chk(_lib.lib().rldm_train_conv_fused(C.byref(d), C.byref(f), _p(x), _p(w_packed), _p(bias), _p(rowadd),
                                          0 if rowadd is None else rowadd.stride(0), _p(res), _p(y), pre, _s(x)),
         "rldm_train_conv_fused")
    return (y, out2) if out2 is not None else y


def wgrad_fused_ok(srcs, N, taps, gn=None):
    d = _desc_srcs(srcs, N, taps, 1, 0)
    f, _ = _fuse(srcs, gn)
    return bool(_lib.lib().rldm_train_wgrad_fused_ok(C.byref(d), C.byref(f)))


def wgrad_fused(dy, srcs, dw, taps, gn=None, rows=None, total=None, rows_accumulate=False):
    """dw += dy (x) act(GN(cat(srcs))) (+ the column sums of dy as wgrad_bias)."""
    d = _desc_srcs(srcs, dy.shape[3], taps, 1, 0)
    f, keep = _fuse(srcs, gn)
    _chk(_lib.lib().rldm_train_wgrad_fused(C.byref(d), C.byref(f), _p(dy), _p(srcs[0].t), _p(dw), _p(rows),
                                           0 if rows is None else rows.stride(0), 1 if rows_accumulate else 0, _p(total), _s(dy)),
         "rldm_train_wgrad_fused")


def chan_stats(x):
    B, W, H, Cc = x.shape
    cs = zeros_stats(B, Cc, x)
    _chk(_lib.lib().rldm_train_chan_stats(_p(x), B, W * H, Cc, _p(cs), _s(x)), "rldm_train_chan_stats")
    return cs


def gn_backward_apply(dz, srcs, gs, gn, dgamma, dbeta, res=None, dsts=None, accumulate=(False, False)):
    """-> the input gradients of GroupNorm(cat(srcs)) (one tensor per source), from dz / gs of a fused data-gradient conv;
    dsts[i] given: written (accumulate[i]: added) in place."""
    B, W, H, Cc = dz.shape
    dsts = list(dsts) if dsts is not None else [None] * len(srcs)
    for i, s in enumerate(srcs):
        if dsts[i] is None:
            dsts[i] = torch.empty_like(s.t)
    two = len(srcs) == 2
    _chk(_lib.lib().rldm_train_gn_backward_apply(_p(dz), _p(srcs[0].t), _p(srcs[1].t) if two else None, srcs[0].t.shape[-1],
                                                 _p(srcs[0].cs), _p(srcs[1].cs) if two else None, _p(gs), B, W * H, Cc, gn.groups,
                                                 gn.eps, _p(gn.gamma), _p(res), _p(dsts[0]), 1 if accumulate[0] else 0,
                                                 _p(dsts[1]) if two else None, 1 if (two and accumulate[1]) else 0, _p(dgamma),
                                                 _p(dbeta), _s(dz)), "rldm_train_gn_backward_apply")
    return dsts


def defer_reduce(on):
    """Weight-gradient partial-tile reductions ride on the next conv launch (rldm_train_defer_reduce); off: flush + launch eagerly."""
    _chk(_lib.lib().rldm_train_defer_reduce(1 if on else 0), "rldm_train_defer_reduce")


def flush_reduce():
    _chk(_lib.lib().rldm_train_flush_reduce(), "rldm_train_flush_reduce")


def wgrad_group(on):
    """Queue the all-taps weight gradients instead of launching them (rldm_train_wgrad_group); off: flush."""
    _chk(_lib.lib().rldm_train_wgrad_group(1 if on else 0), "rldm_train_wgrad_group")


def wgrad_group_flush():
    """Run the queued weight gradients as grouped launches on the stream they were queued on."""
    _chk(_lib.lib().rldm_train_wgrad_group_flush(), "rldm_train_wgrad_group_flush")


def wgrad_group_pending():
    return int(_lib.lib().rldm_train_wgrad_group_pending())
