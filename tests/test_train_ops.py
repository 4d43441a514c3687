"""GPU: the op-level training kernels (rangeldm_amd/csrc/train.hip) against torch fp32 autograd of
the oracle's leaf ops (oracle/ops.py: circular-W / zero-H conv, GroupNorm + SiLU, head_dim-8 attention).

Tolerances: GEMM operands are rounded to bf16 (8 mantissa bits) with fp32 accumulation, so conv / linear results and
gradients carry ~4e-3 relative L2 error against pure fp32 (attention: see its test); everything else (GroupNorm,
elementwise, AdamW) is fp32 end to end: 1e-5 relative.
"""
import numpy as np
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import ops as o_ops

pytestmark = pytest.mark.gpu
TOL_MM = 6e-3
TOL_F32 = 2e-5


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("B,Cin,N,W,H,taps,stride,mode", [
    (2, 32, 64, 16, 8, 9, 1, 0), (3, 48, 32, 8, 4, 9, 1, 0), (2, 5, 32, 16, 8, 9, 1, 0), (2, 32, 4, 16, 8, 9, 1, 0),
    (2, 64, 64, 16, 8, 9, 2, 0), (2, 32, 32, 8, 4, 9, 1, 1), (2, 96, 40, 8, 4, 1, 1, 0), (1, 9, 32, 16, 2, 9, 1, 0),
    (5, 128, 512, 1, 1, 1, 1, 0),                       # a Linear on 5 rows
    # the all-taps weight-gradient kernel (64-multiples of channels, stride 1): every chunk geometry (H = 16 / 8 / 4 / 2),
    # several chunks per workgroup, the nearest-x2 index map, 1x1
    (2, 64, 64, 16, 8, 9, 1, 0), (1, 128, 64, 32, 16, 9, 1, 0), (3, 64, 128, 32, 4, 9, 1, 0), (2, 192, 64, 64, 2, 9, 1, 0),
    (2, 64, 64, 8, 4, 9, 1, 1), (2, 128, 64, 16, 8, 1, 1, 0), (9, 64, 64, 64, 16, 9, 1, 0), (2, 64, 64, 24, 8, 9, 1, 0),
    # launches large enough for the halo-tile conv (no K split): H = 16 (wide tile), 8 / 4 / 2 (narrow tile; 2: three halo passes)
    (4, 64, 128, 256, 16, 9, 1, 0), (8, 128, 128, 128, 8, 9, 1, 0), (16, 64, 256, 64, 4, 9, 1, 0), (32, 64, 256, 64, 2, 9, 1, 0),
])
def test_conv_forward_dgrad_wgrad(B, Cin, N, W, H, taps, stride, mode):
    from rangeldm_amd import train_ops as T
    k = 3 if taps == 9 else 1
    x = rnd(B, Cin, W, H, seed=1).requires_grad_()
    w = (rnd(N, Cin, k, k, seed=2) / (Cin * taps) ** 0.5).requires_grad_()
    bias = rnd(N, seed=3)
    row = rnd(B, N, seed=4)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if mode == 1 else x
    ref = o_ops.circ_conv2d(xin, w, bias, stride, 1 if taps == 9 else 0) + row[:, :, None, None]
    dy = rnd(*ref.shape, seed=5)
    ref.backward(dy)
    wf, wt = T.pack_weights(w.detach().cuda(), taps)
    xd = nhwc(x.detach())
    y = T.conv(xd, wf, N, taps, stride, mode, bias=bias.cuda(), rowadd=row.cuda())
    assert rel(nchw(y), ref.detach()) < TOL_MM
    # residual + accumulate
    y2 = T.conv(xd, wf, N, taps, stride, mode, res=y, out=y.clone(), accumulate=True)
    assert rel(nchw(y2), 3 * ref.detach() - bias[None, :, None, None] - row[:, :, None, None]) < TOL_MM
    # data gradient: the same kernel on dy with the flipped / transposed copy
    dyd = nhwc(dy)
    if stride == 2:
        dx = T.conv(dyd, wt, Cin, taps, 1, 2)                       # zero insertion
    elif mode == 1:
        dx = T.sum2x2(T.conv(dyd, wt, Cin, taps, 1, 0))             # conv at 2x, then fold the nearest-x2
    else:
        dx = T.conv(dyd, wt, Cin, taps, 1, 0)
    assert rel(nchw(dx), x.grad) < TOL_MM
    # weight gradient into a zeroed buffer, twice (accumulates)
    dw = torch.zeros_like(w.detach()).cuda()
    T.wgrad(dyd, xd, dw, taps, stride, mode)
    assert rel(dw.cpu(), w.grad) < TOL_MM
    T.wgrad(dyd, xd, dw, taps, stride, mode)
    assert rel(dw.cpu(), 2 * w.grad) < TOL_MM
    # the same with the bias / row sums from the same pass (taken from the staged bf16 tile where the all-taps kernel applies)
    dw3, rows3, tot3 = torch.zeros_like(dw), torch.full((B, N + 4), 7.0).cuda(), torch.zeros(N).cuda()
    T.wgrad_bias(dyd, xd, dw3, taps, stride, mode, rows=rows3[:, 4:], total=tot3)
    assert rel(dw3.cpu(), w.grad) < TOL_MM and float((rows3[:, :4] - 7.0).abs().max()) == 0
    assert rel(rows3[:, 4:].cpu(), dy.sum((2, 3))) < 3e-3 and rel(tot3.cpu(), dy.sum((0, 2, 3))) < 3e-3
    # bias / per-sample row gradients
    rows = torch.zeros(B, N).cuda()
    tot = torch.zeros(N).cuda()
    T.colsum(dyd, rows=rows, total=tot)
    assert rel(rows.cpu(), dy.sum((2, 3))) < TOL_F32 and rel(tot.cpu(), dy.sum((0, 2, 3))) < TOL_F32
    # (round 6) the same weight gradients QUEUED and run by the grouped launch (rldm_train_wgrad_group: shapes the all-taps kernel
    # covers are queued, the others launch at once): two layers' worth in one flush, into a zeroed and into a pre-filled buffer
    dw4, dw5 = torch.zeros_like(dw), dw3.clone()
    rows4, tot4 = torch.zeros(B, N).cuda(), torch.zeros(N).cuda()
    T.wgrad_group(True)
    try:
        T.wgrad_bias(dyd, xd, dw4, taps, stride, mode, rows=rows4, total=tot4)
        T.wgrad(dyd, xd, dw5, taps, stride, mode)
        queued = T.wgrad_group_pending()
        assert queued in (0, 2)
    finally:
        T.wgrad_group(False)
    assert T.wgrad_group_pending() == 0
    assert rel(dw4.cpu(), w.grad) < TOL_MM and rel(dw5.cpu(), 2 * w.grad) < TOL_MM
    assert rel(rows4.cpu(), dy.sum((2, 3))) < 3e-3 and rel(tot4.cpu(), dy.sum((0, 2, 3))) < 3e-3
    if queued:                                      # same kernel body, same operands: the K-slice boundaries differ, nothing else
        assert rel(dw4, dw3) < 1e-4


@pytest.mark.parametrize("B,C,W,H,silu", [(2, 64, 16, 8, True), (3, 32, 8, 4, False), (1, 128, 4, 2, True), (2, 512, 4, 2, True),
                                            (2, 96, 8, 4, True), (2, 160, 4, 2, False), (2, 256, 16, 8, True),       # 3 / 5 / 8 channels per group
                                            (2, 128, 256, 16, True), (1, 512, 64, 8, False)])  # slab kernels (large tensors)
def test_group_norm_forward_backward(B, C, W, H, silu):
    from rangeldm_amd import train_ops as T
    x = (rnd(B, C, W, H, seed=1) * 2 + 0.5).requires_grad_()
    g = (1 + 0.3 * rnd(C, seed=2)).requires_grad_()
    b = (0.2 * rnd(C, seed=3)).requires_grad_()
    ref = o_ops.group_norm_silu(x, g, b, 32, 1e-5, silu)
    dy = rnd(B, C, W, H, seed=4)
    ref.backward(dy)
    xd = nhwc(x.detach())
    y, stats = T.gn_forward(xd, g.detach().cuda(), b.detach().cuda(), 32, 1e-5, silu)
    assert rel(nchw(y), ref.detach()) < TOL_F32
    dg, db = torch.zeros(C).cuda(), torch.zeros(C).cuda()
    dx = T.gn_backward(xd, nhwc(dy), stats, g.detach().cuda(), b.detach().cuda(), 32, silu, dg, db)
    assert rel(nchw(dx), x.grad) < 1e-4
    assert rel(dg.cpu(), g.grad) < 1e-4 and rel(db.cpu(), b.grad) < 1e-4
    dx2 = T.gn_backward(xd, nhwc(dy), stats, g.detach().cuda(), b.detach().cuda(), 32, silu, dg, db, dx=dx.clone(), accumulate=True)
    assert rel(nchw(dx2), 2 * x.grad) < 1e-4


@pytest.mark.parametrize("B,L,C", [(2, 64, 32), (1, 200, 16), (2, 1024, 16), (3, 16, 64), (2, 4, 8), (1, 256, 256)])
def test_attention_forward_backward(B, L, C):
    """MFMA attention (train_attn.hip): q, k, v, dO and the probabilities enter the matrix cores as bf16 (the reference's
    `mixed_precision: bf16`), statistics / accumulation fp32.  Gates: 5e-3 (output) / 8e-3 (gradients: dS is rounded too) relative L2 against fp32 SDPA on bf16-rounded
    inputs (what remains is the rounding of q * scale and of the probabilities), 1.5e-2 against SDPA on the fp32 inputs;
    log-sum-exp 8e-2 absolute (scores reach +-10 here and bf16(q * scale) carries 2^-9 of that)."""
    from rangeldm_amd import train_ops as T
    dO = rnd(B, L, C, seed=4)
    nh = C // 8
    refs = []
    for rounded in (True, False):
        q, k, v = (rnd(B, L, C, seed=s) * 1.5 for s in (1, 2, 3))
        if rounded:
            q, k, v = (z.bfloat16().float() for z in (q, k, v))
        q, k, v = (z.requires_grad_() for z in (q, k, v))
        qh, kh, vh = (z.view(B, L, nh, 8).transpose(1, 2) for z in (q, k, v))
        ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, L, C)
        ref.backward(dO.bfloat16().float() if rounded else dO)
        lse_ref = torch.logsumexp(qh.detach() @ kh.detach().transpose(-1, -2) / math.sqrt(8), -1)
        refs.append((ref.detach(), q.grad, k.grad, v.grad, lse_ref))
    qd, kd, vd = (rnd(B, L, C, seed=s).cuda() * 1.5 for s in (1, 2, 3))
    o, lse = T.attention_forward(qd, kd, vd)
    dq, dk, dv = T.attention_backward(qd, kd, vd, o, dO.cuda(), lse)
    for (ref, gq, gk, gv, lse_ref), tol in zip(refs, (5e-3, 1.5e-2)):
        assert rel(o, ref) < tol, (tol, rel(o, ref))
        assert float((lse.cpu() - lse_ref).abs().max()) < 8e-2
        gt = max(tol, 8e-3)
        assert rel(dq, gq) < gt and rel(dk, gk) < gt and rel(dv, gv) < gt, (gt, rel(dq, gq), rel(dk, gk), rel(dv, gv))


@pytest.mark.parametrize("B,K,N", [(8, 512, 4352), (2, 128, 96), (16, 4352, 512), (3, 40, 24)])
def test_linear_rows_forward_backward(B, K, N):
    """Row-wise Linear kernels (TimestepEmbedding MLP, the fused time_emb_proj of all resnets): weights bf16, rows fp32."""
    from rangeldm_amd import train_ops as T
    x = rnd(B, K, seed=1).requires_grad_()
    w = (rnd(N, K, seed=2) / K ** 0.5).requires_grad_()
    bias = rnd(N, seed=3).requires_grad_()
    ref = F.linear(x, w, bias)
    dy = rnd(B, N, seed=4)
    ref.backward(dy)
    wf, wt = T.pack_weights(w.detach().cuda(), 1)
    wide = torch.zeros(B, K + 8).cuda()                 # rows as a column slice of a wider matrix
    wide[:, 4:4 + K] = x.detach().cuda()
    y = T.linear_rows(wide[:, 4:4 + K], wf, N, bias=bias.detach().cuda())
    assert rel(y, ref.detach()) < TOL_MM
    dyd = dy.cuda()
    assert rel(T.linear_rows(dyd, wt, K), x.grad) < TOL_MM
    dw, db = torch.zeros(N, K).cuda(), torch.zeros(N).cuda()
    for _ in range(2):                                  # accumulates
        T.linear_rows_wgrad(dyd, wide[:, 4:4 + K], dw, db)
    assert rel(dw, 2 * w.grad) < 1e-5 and rel(db, 2 * bias.grad) < 1e-5


def test_elementwise_and_loss():
    from rangeldm_amd import train_ops as T
    a, b = rnd(2, 8, 4, 6, seed=1).cuda(), rnd(2, 8, 4, 6, seed=2).cuda()
    assert torch.equal(T.add(a, b), a + b)
    c = T.concat(a, b[..., :3].contiguous())
    assert torch.equal(c, torch.cat([a, b[..., :3]], -1))
    d = torch.zeros_like(a)
    T.copy_channels(c, 2, d, 1, 4)
    T.copy_channels(c, 2, d, 1, 4, accumulate=True)
    assert torch.equal(d[..., 1:5], 2 * c[..., 2:6]) and float(d[..., 0].abs().max()) == 0
    du = rnd(2, 8, 4, 6, seed=3).cuda()
    ref = du.view(2, 4, 2, 2, 2, 6).sum((2, 4))
    assert rel(T.sum2x2(du), ref) < 1e-6
    z = rnd(5, 512, seed=4).requires_grad_()
    F.silu(z).backward(torch.ones_like(z) * 0.5)
    assert rel(T.silu(z.detach().cuda()), F.silu(z).detach()) < 1e-6
    assert rel(T.silu_backward(z.detach().cuda(), torch.full((5, 512), 0.5).cuda()), z.grad) < 1e-5
    t = torch.tensor([0, 3, 977, 500, 999])
    assert rel(T.timestep_embedding(t.cuda(), 128), o_ops.timestep_embedding(t, 128)) < 2e-5
    x = rnd(2, 4, 8, 4, seed=5)
    pk = T.pack_input(x.cuda(), True)
    pe = torch.zeros(2, 1, 8, 4)
    pe[:, :, 0, :] = 1
    assert torch.equal(pk.cpu(), torch.cat([x, pe], 1).permute(0, 2, 3, 1))
    assert torch.equal(T.unpack_output(pk).cpu(), torch.cat([x, pe], 1))
    pred = rnd(3, 4, 8, 4, seed=6).requires_grad_()
    tgt = rnd(3, 4, 8, 4, seed=7)
    wts = torch.tensor([0.5, 1.0, 0.25])
    loss = (F.mse_loss(pred, tgt, reduction="none").mean((1, 2, 3)) * wts).mean()
    loss.backward()
    l, dp = T.mse(nhwc(pred.detach()), tgt.cuda(), wts.cuda())
    assert abs(float(l) - float(loss)) < 1e-6 and rel(nchw(dp), pred.grad) < 1e-6
    l2, _ = T.mse(nhwc(pred.detach()), tgt.cuda())
    assert abs(float(l2) - float(F.mse_loss(pred, tgt))) < 1e-6


def test_adamw_clip_ema_match_torch():
    """torch.optim.AdamW(lr, betas (0.95, 0.999), wd 1e-6, eps 1e-8) + clip_grad_norm_(1.0) + EMAModel.step
    (ldm/train_unconditional.py:357-363,548,556) on a flat buffer."""
    from rangeldm_amd import train_ops as T
    n = 10007
    p0 = rnd(n, seed=1)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=1e-3, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    ema_ref = p0.clone()
    pd, m, v, ema = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda(), p0.clone().cuda()
    for step in range(1, 4):
        g = rnd(n, seed=10 + step) * (3.0 if step == 2 else 0.001)        # step 2 clips, the others do not
        p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([p], 1.0)
        opt.step()
        decay = 0.9 + 0.01 * step
        ema_ref.sub_((1 - decay) * (ema_ref - p.detach()))
        gd = g.cuda()
        sq = T.sqnorm(gd)
        assert abs(float(sq) - float((g.double() ** 2).sum())) < 1e-6 * float((g.double() ** 2).sum())
        T.adamw(pd, gd, m, v, step, 1e-3, (0.95, 0.999), 1e-8, 1e-6, ema=ema, ema_decay=decay, sqnorm_dev=sq, max_grad_norm=1.0)
        assert rel(pd, p.detach()) < 1e-6 and rel(ema, ema_ref) < 1e-6


# ---- fused tape (round 5): GroupNorm folded into its consumers / producers ------------------------------------------------
@pytest.mark.parametrize("B,C0,C1,N,W,H,taps,silu", [
    (8, 256, 0, 256, 32, 2, 9, True),        # 32x2 level: split-K launch, the tile's last arriver runs the epilogue
    (8, 256, 128, 256, 64, 4, 9, True),      # two sources, 12 channels per group (groups straddle the seam), split-K
    (4, 128, 0, 128, 256, 16, 9, True),      # halo-tile kernel, 64-channel tiles
    (8, 128, 0, 128, 256, 16, 9, True),      # halo-tile kernel, 128-channel tiles
    (2, 128, 128, 384, 128, 8, 1, False),    # 1x1 (to_q/k/v form: no SiLU), two sources, no split
    (2, 96, 0, 64, 32, 4, 9, True),          # 32-channel chunks, 3 channels per group
    (2, 512, 256, 256, 32, 2, 1, True),      # 768 input channels (24 per group)
    (3, 64, 64, 64, 64, 2, 9, True),         # three halo passes, two sources
])
def test_fused_norm_conv_forward_backward(B, C0, C1, N, W, H, taps, silu):
    """act(GN(cat(x0, x1))) -> conv + bias + row + res in ONE launch with the output's statistics; backward: the data gradient with
    the GroupNorm-backward transform in its epilogue + rldm_train_gn_backward_apply; the weight gradient rebuilding its operand."""
    from rangeldm_amd import train_ops as T
    k = 3 if taps == 9 else 1
    Cin = C0 + C1
    x = (rnd(B, Cin, W, H, seed=1) * 1.5 + 0.3).requires_grad_()
    g = (1 + 0.3 * rnd(Cin, seed=2)).requires_grad_()
    b = (0.2 * rnd(Cin, seed=3)).requires_grad_()
    w = (rnd(N, Cin, k, k, seed=4) / (Cin * taps) ** 0.5).requires_grad_()
    bias, row, res = rnd(N, seed=5), rnd(B, N, seed=6), rnd(B, N, W, H, seed=7)
    h = o_ops.group_norm_silu(x, g, b, 32, 1e-5, silu)
    ref = o_ops.circ_conv2d(h, w, bias, 1, 1 if taps == 9 else 0) + row[:, :, None, None] + res
    dy = rnd(*ref.shape, seed=8)
    ref.backward(dy)
    xd = nhwc(x.detach())
    srcs = [T.Src(xd[..., :C0].contiguous())] + ([T.Src(xd[..., C0:].contiguous())] if C1 else [])
    for s in srcs:
        s.cs = T.chan_stats(s.t)
        xs = nchw(s.t)
        assert rel(s.cs[..., 0].cpu(), xs.sum((2, 3))) < TOL_F32 and rel(s.cs[..., 1].cpu(), (xs * xs).sum((2, 3))) < TOL_F32
    gn = T.GN(g.detach().cuda(), b.detach().cuda(), silu, 32, 1e-5)
    wf, wt = T.pack_weights(w.detach().cuda(), taps)
    assert T.conv_fused_ok(srcs, N, taps, gn=gn, want_stats=True)
    y, cs = T.conv_fused(srcs, wf, N, taps, gn=gn, bias=bias.cuda(), rowadd=row.cuda(), res=nhwc(res), want_stats=True)
    assert rel(nchw(y), ref.detach()) < TOL_MM
    yr = nchw(y)
    assert rel(cs[..., 0].cpu(), yr.sum((2, 3))) < 1e-4 and rel(cs[..., 1].cpu(), (yr * yr).sum((2, 3))) < 1e-4
    # the same without the statistics epilogue (register epilogue of the fused instance)
    y1 = T.conv_fused(srcs, wf, N, taps, gn=gn, bias=bias.cuda(), rowadd=row.cuda(), res=nhwc(res))
    assert rel(nchw(y1), ref.detach()) < TOL_MM
    # backward: d act(GN(x)) = conv^T(dy), transformed to dz in the epilogue
    dyd = nhwc(dy)
    dsrc = [T.Src(dyd)]
    assert T.conv_fused_ok(dsrc, Cin, taps, gsrcs=srcs, ggn=gn)
    dz, gs = T.conv_fused(dsrc, wt, Cin, taps, gsrcs=srcs, ggn=gn)
    dg, db = torch.zeros(Cin).cuda(), torch.zeros(Cin).cuda()
    extra = nhwc(rnd(B, Cin, W, H, seed=9))                       # a residual-path gradient added by the same launch
    pre1 = torch.full_like(srcs[-1].t, 0.5)
    dsts = T.gn_backward_apply(dz, srcs, gs, gn, dg, db, res=extra, dsts=[None, pre1][:len(srcs)], accumulate=(False, True))
    want = x.grad + nchw(extra)
    assert rel(nchw(dsts[0]), want[:, :C0]) < TOL_MM
    if C1:
        assert rel(nchw(dsts[1]), want[:, C0:] + 0.5) < TOL_MM
    assert rel(dg.cpu(), g.grad) < TOL_MM and rel(db.cpu(), b.grad) < TOL_MM
    # weight gradient with the operand rebuilt from the raw sources
    if T.wgrad_fused_ok(srcs, N, taps, gn=gn):
        dw, rows, tot = torch.zeros_like(w.detach()).cuda(), torch.zeros(B, N).cuda(), torch.zeros(N).cuda()
        T.wgrad_fused(dyd, srcs, dw, taps, gn=gn, rows=rows, total=tot)
        assert rel(dw.cpu(), w.grad) < TOL_MM
        assert rel(rows.cpu(), dy.sum((2, 3))) < 3e-3 and rel(tot.cpu(), dy.sum((0, 2, 3))) < 3e-3
        # (round 6) queued + grouped launch (the V3 staging where the image has 8 / 16 beams, the round-5 staging below)
        dw2, rows2, tot2 = torch.zeros_like(dw), torch.zeros(B, N).cuda(), torch.zeros(N).cuda()
        T.wgrad_group(True)
        try:
            T.wgrad_fused(dyd, srcs, dw2, taps, gn=gn, rows=rows2, total=tot2)
            assert T.wgrad_group_pending() == 1
        finally:
            T.wgrad_group(False)
        assert rel(dw2.cpu(), w.grad) < TOL_MM and rel(dw2, dw) < 2e-3
        assert rel(rows2.cpu(), dy.sum((2, 3))) < 3e-3 and rel(tot2.cpu(), dy.sum((0, 2, 3))) < 3e-3
    else:
        assert Cin % 64 != 0
