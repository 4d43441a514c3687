"""GPU parity, kernel level: each HIP kernel against the CPU oracle's leaf ops on identical seeded inputs.

Tolerances (stated per BASELINE.md section 4): the kernels read bf16 activations/weights and accumulate in fp32, so
  * against the oracle evaluated on the SAME bf16-rounded operands  -> rel-L2 <= 4e-3 (one bf16 rounding of the output)
  * against the pure fp32 oracle                                     -> rel-L2 <= 2e-2
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import ops
from tests.hip_util import bf16r, rel_l2, hip_conv, hip_attention, hip_attention_qkv, hip_conv_stats

pytestmark = pytest.mark.gpu
TOL_Q, TOL_F = 4e-3, 2e-2


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CONV_CASES = [
    # (B, Cin, Cout, W, H, k, stride, pad_mode, up)       -- shapes from SURVEY.md A.4 (reduced batch) + edge cases
    (2, 128, 128, 32, 16, 3, 1, 0, False),     # L0-like, CK=64
    (1, 128, 128, 256, 16, 3, 1, 0, False),    # full L0 width (wrap seam across many tiles)
    (2, 256, 256, 32, 2, 3, 1, 0, False),      # L3: H=2, mostly zero padding
    (4, 256, 256, 32, 1, 3, 1, 0, False),      # nuScenes L3: 32x1 images (conv_small.hip's 32-pixel tiles), H=1: only zero rows above / below
    (3, 512, 256, 32, 1, 3, 1, 0, False),      # ... its up-block input width
    (2, 256, 256, 32, 1, 1, 1, 0, False),      # ... pointwise
    (2, 256, 256, 64, 4, 3, 1, 0, False),      # L2
    (3, 128, 256, 16, 8, 3, 1, 0, False),      # odd batch
    (2, 16, 128, 32, 16, 3, 1, 0, False),      # conv_in (padded 16-ch input), CK=16
    (2, 5, 128, 32, 16, 3, 1, 0, False),       # conv_in with the real 5 channels
    (2, 128, 4, 32, 16, 3, 1, 0, False),       # conv_out-like N=4 (BN=32, masked)
    (2, 64, 2, 64, 32, 3, 1, 0, False),        # VAE conv_out N=2
    (2, 128, 128, 32, 16, 3, 2, 0, False),     # UNet downsample s2 p1
    (2, 64, 64, 64, 32, 3, 2, 1, False),       # VAE downsample s2 end-pad
    (2, 128, 128, 16, 8, 3, 1, 0, True),       # nearest x2 folded
    (1, 256, 256, 32, 16, 3, 1, 0, True),      # VAE decoder up conv
    (2, 128, 256, 32, 8, 1, 1, 0, False),      # 1x1 shortcut
    (2, 32, 32, 32, 8, 3, 1, 0, False),        # small-config channels (CK=16 path)
    (2, 32, 64, 8, 1, 3, 1, 0, False),         # H=1 (nuScenes deepest level)
    (1, 64, 64, 1024, 64, 3, 1, 0, False),     # VAE full resolution
    (2, 512, 256, 32, 2, 3, 1, 0, False),      # conv_small.hip: L3 up-block conv1 (concat width), 32-channel tiles
    (1, 384, 256, 64, 4, 3, 1, 0, False),      # conv_small.hip: L2, Cin = 256 + 128
    (16, 256, 256, 64, 4, 3, 1, 0, False),     # conv_small.hip: L2 at the bench batch -> 64-channel tiles
    (2, 128, 64, 32, 8, 3, 1, 0, False),       # conv_small.hip: 8x8 pixel tiles
    (3, 64, 64, 16, 4, 3, 1, 0, False),        # conv_small.hip: one 64-pixel image per block, Cin = 64 (64-channel tiles only)
    (2, 128, 128, 128, 8, 3, 1, 0, False),     # conv_small.hip: L1, 128-pixel tiles (16x8)
    (16, 256, 128, 128, 8, 3, 1, 0, False),    # conv_small.hip: L1 at the bench batch (one full round of workgroups)
    (1, 384, 128, 128, 8, 3, 1, 0, False),     # conv_small.hip: L1 up-block width 256 + 128
    (3, 256, 256, 32, 2, 3, 1, 0, True),       # conv_small.hip: nearest x2 folded into the staging (L3 upsample)
    (16, 128, 128, 256, 16, 3, 1, 0, False),   # conv_stream.hip: L0 at the bench batch (256 workgroups)
    (1, 256, 256, 64, 16, 3, 1, 0, False),     # conv_stream.hip (flag): 4 chunks, two channel tiles
    (1, 64, 128, 32, 8, 3, 1, 0, False),       # conv_stream.hip (flag): a single chunk, a single tile
    (1, 128, 128, 32, 16, 3, 1, 0, True),      # conv_stream.hip (flag): nearest x2 folded (64x32 output)
    (32, 128, 128, 128, 4, 3, 1, 0, False),    # conv_stream.hip: images of 4 beams (nuScenes 128x4 level at batch 32): 32 x 4 pixel tiles
    (2, 256, 128, 1024, 4, 3, 1, 0, False),    # ... wide image, 4 chunks
    (2, 128, 64, 1024, 64, 3, 1, 0, False),    # conv_stream.hip <2, 2>: 64 output channels on 256-pixel tiles, 2 k-groups (VAE decoder, 128 -> 64)
    (1, 128, 64, 512, 32, 3, 1, 0, True),      # ... behind the folded nearest x2 (the decoder's last upsample is 128 -> 128; this is the shape check)
    (2, 64, 192, 64, 16, 3, 1, 0, False),      # ... (flag) three 64-channel tiles
    (16, 256, 256, 128, 8, 3, 1, 0, False),    # conv_stream.hip, 4-wave 128 x 64 x 2 k-groups: the 128x8 level at the bench batch (512 workgroups, two per CU)
    (3, 128, 128, 256, 16, 3, 1, 0, False),    # ... 4-wave 128 x 128 under "stream-any-grid" (96 workgroups), odd batch
    (16, 128, 128, 256, 16, 3, 2, 0, False),   # conv_stream.hip, stride 2 on 64-pixel x 128-channel tiles (17 x 17 halo): the 256x16 -> 128x8 down-sampler at the bench batch
    (3, 256, 128, 64, 32, 3, 2, 0, False),     # ... four input chunks, odd batch (under "stream-any-grid")
    (16, 128, 128, 128, 8, 3, 2, 0, False),    # ... 16 x 4 tiles for an output of 4 beams (the 128x8 -> 64x4 down-sampler at the bench batch)
    (16, 128, 128, 128, 8, 3, 1, 0, True),     # conv_stream.hip, sub-pixel form of nearest x2 + 3x3 (four 2x2 convs over the input): the 128x8 -> 256x16 up-sampler at the bench batch
    (2, 256, 256, 256, 16, 3, 1, 0, True),     # ... the VAE decoder's first up-sampler (4 chunks, two channel tiles)
    (3, 128, 256, 64, 8, 3, 1, 0, True),       # ... odd batch, one tile row (zero rows above and below every tile)
    (16, 256, 256, 64, 4, 3, 1, 0, True),      # ... inputs of 4 beams on 32 x 4 tiles: the 64x4 -> 128x8 up-sampler at the bench batch
    (2, 128, 128, 128, 4, 3, 1, 0, True),      # ... (flag) nuScenes' 128x4 -> 256x8
    (16, 5, 128, 256, 16, 3, 1, 0, False),     # conv_regw.hip, conv_c16_kernel: the UNet's conv_in at the bench batch (5 real input channels of 16)
    (3, 4, 256, 256, 16, 3, 1, 0, False),      # ... the decoder's conv_in (two 128-channel groups, odd batch)
    (1, 16, 128, 1024, 8, 3, 1, 0, False),     # ... all 16 channels real, one tile row, 64 tiles across the wrap
    (16, 256, 256, 64, 4, 3, 2, 0, False),     # conv_regw.hip, conv_ds2_kernel: the 64x4 -> 32x2 down-sampler at the bench batch (8 waves = 8 k-groups)
    (3, 256, 64, 128, 2, 3, 2, 0, False),      # ... one output row (zero rows above AND below), two tiles across the wrap, odd batch
    (4, 256, 256, 64, 8, 3, 2, 0, False),      # ... four output rows
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_geometry(case, conv_flags):
    B, Cin, Cout, W, H, k, s, pm, up = case
    x = _rand(B, Cin, W, H, seed=1)
    w = _rand(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    b = _rand(Cout, seed=3, scale=0.1)
    y = hip_conv(x, w, b, stride=s, pad_mode=pm, upsample=up)
    xq, wq = bf16r(x), bf16r(w)

    def ref(xx, ww):
        if up:
            return ops.upsample_conv(xx, ww, b)
        if s == 2 and pm == 1:
            return ops.downsample_vae(xx, ww, b)
        return ops.circ_conv2d(xx, ww, b, s, 1 if k == 3 else 0)

    assert y.shape == ref(xq, wq).shape
    assert rel_l2(y, ref(xq, wq)) < TOL_Q
    assert rel_l2(y, ref(x, w)) < TOL_F


def test_conv_wrap_seam_exact():
    # an impulse at azimuth 0 must leak to azimuth W-1 (wrap) and an impulse at beam 0 must NOT leak to beam H-1 (zero pad)
    W, H, C = 32, 16, 16
    x = torch.zeros(1, C, W, H)
    x[0, 0, 0, 0] = 1.0
    w = torch.zeros(32, C, 3, 3)
    w[0, 0, 0, 1] = 1.0     # tap (i=0: reads w-1, j=1: same beam)  -> out(w) = in(w-1)
    w[1, 0, 2, 1] = 1.0     # reads w+1                               -> out(w) = in(w+1)
    w[2, 0, 1, 2] = 1.0     # reads h+1                               -> out(h) = in(h+1)
    y = hip_conv(x, w, torch.zeros(32))
    ref = ops.circ_conv2d(x, w, torch.zeros(32))
    assert torch.equal(y, ref)
    assert y[0, 0, 1, 0] == 1 and y[0, 1, W - 1, 0] == 1        # wrap both ways
    assert y[0, 2].abs().sum() == 0                             # beam -1 does not exist: zero padding


GN_CASES = [(256, 256, 256, 32, 1), (128, 128, 128, 32, 16), (256, 128, 256, 16, 8), (256, 256, 256, 32, 2), (64, 32, 64, 16, 8),
            (256, 128, 256, 64, 4), (128, 128, 128, 16, 4), (256, 128, 128, 128, 8), (128, 128, 128, 128, 8),
            (128, 64, 128, 64, 16), (64, 64, 256, 32, 8), (64, 64, 64, 32, 8), (128, 64, 64, 16, 16),
            (128, 128, 128, 1024, 4),       # (conv_stream.hip on 32 x 4 tiles: concat, GroupNorm, time embedding, residual)
            (32, 32, 64, 1024, 64), (64, 64, 64, 256, 16),
            (256, 256, 192, 32, 8), (256, 256, 128, 64, 8)]  # (4-wave conv_stream instances, 512 input channels: two channels per thread in the GroupNorm fold)   # (conv_stream.hip <2, 2>: 64 output channels, identity residual of 64)


@pytest.fixture(params=[0, 1024, 4096, (4096, 7), (4096, 7 + 32), (4096, 64), 256 + 2048, 524288, 1 << 22],
                ids=["default", "small-128px-tiles", "stream-any-grid", "stream-any-grid-8-waves", "stream-any-grid-specialised-waves",
                     "stream-any-grid-64px-tiles", "generic-only", "level3-64px-tiles", "own-image-tiles"])
def conv_flags(request):
    """Routing of the conv launches: 0 default; 1024 also sends the 128x8 level to conv_small.hip (128-pixel tiles);
    4096 sends every eligible 3x3 to conv_stream.hip regardless of the grid size (by default it needs >= 128 workgroups);
    256 + 2048 keeps everything on the generic implicit-GEMM kernel; 524288 keeps 32x2 images on one 64-pixel conv_small tile
    (by default they run as two 32-pixel tiles); 1 << 22 makes every conv that can own a whole image (<= 64 pixels, conv_small.hip)
    do so and write two normalised copies of its output (producer-side GroupNorm epilogue; the copies are checked at network level:
    test_producer_side_groupnorm_matches_consumer_side)."""
    from rangeldm_amd import _lib
    f1, f2 = request.param if isinstance(request.param, tuple) else (request.param, 0)
    _lib.lib().rldm_debug_set_flags(f1)
    _lib.lib().rldm_debug_set_flags2(f2)      # (7: round 4's 4-wave conv_stream workgroups off -> the 8-wave instances keep their coverage;
                                              #  32: the 256 x 128 tile runs with specialised matrix / staging waves;
                                              #  64: 8 x 8 tiles x 128 channels x 2 k-groups wherever 128 | N -- the 128x8 level's default)
    yield f1
    _lib.lib().rldm_debug_set_flags(0)
    _lib.lib().rldm_debug_set_flags2(0)


@pytest.mark.parametrize("C0,C1,Cout,W,H", GN_CASES)
def test_conv_gn_silu_concat_temb_residual(C0, C1, Cout, W, H, conv_flags):
    """The full ResnetBlock conv1 fusion: GN(32)+SiLU over cat[x0,x1] -> conv3x3 -> +bias +temb[b] (+res)."""
    B = 2
    x0, x1 = _rand(B, C0, W, H, seed=4) * 1.5 + 0.3, _rand(B, C1, W, H, seed=5) * 0.7 - 0.2
    Cin = C0 + C1
    w = _rand(Cout, Cin, 3, 3, seed=6, scale=(Cin * 9) ** -0.5)
    b = _rand(Cout, seed=7, scale=0.1)
    gamma, beta = 1 + 0.2 * _rand(Cin, seed=8), 0.2 * _rand(Cin, seed=9)
    temb = _rand(B, Cout, seed=10)
    res = _rand(B, Cout, W, H, seed=11)
    y = hip_conv(x0, w, b, x1=x1, gamma=gamma, beta=beta, silu=True, eps=1e-5, temb=temb, res=res)

    def ref(q):
        xc = torch.cat([q(x0), q(x1)], 1)
        h = q(ops.group_norm_silu(xc, gamma, beta, 32, 1e-5))
        return ops.circ_conv2d(h, q(w), b) + temb[:, :, None, None] + q(res)

    assert rel_l2(y, ref(bf16r)) < TOL_Q
    assert rel_l2(y, ref(lambda t: t)) < TOL_F


@pytest.fixture(params=[0, 1 << 25, 1 << 24], ids=["default", "runs-of-8-workgroups", "per-tile-kernel"])
def regw_flags(request):
    """conv_regw.hip (64 -> 64 channels, weights in registers, a run of 16 x 8 tiles per workgroup): by default where an image batch has at
    least 2048 tiles; 1 << 25 caps the grid at 8 workgroups so that small images are walked in runs too; 1 << 24 keeps conv_stream's
    per-tile instance (the comparison)."""
    from rangeldm_amd import _lib
    _lib.lib().rldm_debug_set_flags2(request.param)
    yield request.param
    _lib.lib().rldm_debug_set_flags2(0)


@pytest.mark.parametrize("B,W,H,gn,res", [(4, 512, 64, True, False), (4, 512, 64, True, True), (2, 128, 16, True, True), (2, 128, 16, False, False),
                                          (3, 64, 32, True, True), (1, 256, 8, True, False), (2, 1024, 64, False, True)])
def test_conv_c64_register_weights(B, W, H, gn, res, regw_flags):
    """The VAE decoder's full-resolution ResnetBlock convs (64 -> 64): GroupNorm(32) + SiLU -> conv3x3 -> + bias (+ x)."""
    x = _rand(B, 64, W, H, seed=50) * 1.3 + 0.2
    w = _rand(64, 64, 3, 3, seed=51, scale=(64 * 9) ** -0.5)
    b = _rand(64, seed=52, scale=0.1)
    gamma, beta = (1 + 0.2 * _rand(64, seed=53), 0.2 * _rand(64, seed=54)) if gn else (None, None)
    r = _rand(B, 64, W, H, seed=55) if res else None
    y = hip_conv(x, w, b, gamma=gamma, beta=beta, silu=gn, eps=1e-6, res=r)

    def ref(q):
        h = q(ops.group_norm_silu(q(x), gamma, beta, 32, 1e-6)) if gn else q(x)
        out = ops.circ_conv2d(h, q(w), b)
        return out + q(r) if res else out

    assert rel_l2(y, ref(bf16r)) < TOL_Q
    assert rel_l2(y, ref(lambda t: t)) < TOL_F
    # per image and channel: a wrong tile of a run, a stale halo buffer or a lost residual is a local error the L2 norm forgives
    d = (y - ref(bf16r)).abs().amax(dim=(2, 3))
    assert float(d.max()) < 0.05


@pytest.mark.parametrize("B,N,W,H", [(16, 256, 64, 4), (3, 64, 128, 2)])
def test_conv_stride2_kgroup_waves_statistics(B, N, W, H):
    """conv_regw.hip's conv_ds2_kernel: the (sum, sumsq) side output of the stride-2 down-sampler (one partial per 32 x 1 output tile)."""
    x = _rand(B, 256, W, H, seed=90)
    w = _rand(N, 256, 3, 3, seed=91, scale=(256 * 9) ** -0.5)
    b = _rand(N, seed=92, scale=0.5)
    y = hip_conv(x, w, b, stride=2)
    st = hip_conv_stats(x, w, b, stride=2)
    ref_s = y.double().sum(dim=(2, 3))
    ref_q = (y.double() ** 2).sum(dim=(2, 3))
    assert float((st[..., 0].double() - ref_s).abs().max()) < 1e-4 * W * H
    assert float(((st[..., 1].double() - ref_q).abs() / (ref_q + 1e-6)).max()) < 1e-4


def test_conv_c64_register_weights_ping_pong_variant():
    """RLDM_RW_TEAMS=2 (read once per process: a child process): conv_regw.hip as ONE 8-wave workgroup per CU whose two teams swap roles at
    barriers -- the measured alternative of DESIGN.md section 3.11 stays a working switch."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, RLDM_RW_TEAMS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-k",
                        "test_conv_c64_register_weights and not ping_pong and (512 or 128)", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("B,W,H,N", [(4, 512, 64, 2), (2, 128, 16, 2), (3, 64, 32, 4), (2, 1024, 64, 2), (2, 128, 16, 1)])
@pytest.mark.parametrize("flags", [1 << 26, (1 << 26) | (1 << 25), (1 << 26) | (1 << 24)], ids=["default", "runs-of-8-workgroups", "generic-kernel"])
def test_conv_out_fp32_nchw(B, W, H, N, flags):
    """The VAE decoder's output layer: GroupNorm(32) + SiLU -> conv3x3 (64 -> 2) written as fp32 NCHW by the kernel itself
    (rldm_debug_set_flags2(1 << 26) makes the test conv such an output layer): conv_regw.hip's one-tile variant, or the generic kernel."""
    from rangeldm_amd import _lib
    x = _rand(B, 64, W, H, seed=70) * 1.3 + 0.2
    w = _rand(N, 64, 3, 3, seed=71, scale=(64 * 9) ** -0.5)
    b = _rand(N, seed=72, scale=0.1)
    gamma, beta = 1 + 0.2 * _rand(64, seed=73), 0.2 * _rand(64, seed=74)
    _lib.lib().rldm_debug_set_flags2(flags)
    try:
        y = hip_conv(x, w, b, gamma=gamma, beta=beta, silu=True, eps=1e-6)
    finally:
        _lib.lib().rldm_debug_set_flags2(0)

    def ref(q):
        return ops.circ_conv2d(q(ops.group_norm_silu(q(x), gamma, beta, 32, 1e-6)), q(w), b)

    # fp32 outputs (no bf16 rounding of the result): the quantised reference is met to accumulation order
    assert rel_l2(y, ref(bf16r)) < 2e-3
    assert rel_l2(y, ref(lambda t: t)) < TOL_F
    assert float((y - ref(bf16r)).abs().max()) < 0.02


@pytest.mark.parametrize("B,W,H,N", [(16, 256, 16, 4), (3, 256, 16, 2), (2, 1024, 8, 4), (1, 1024, 64, 2)])
@pytest.mark.parametrize("route", ["conv_o4", "generic"])
def test_unet_output_layer_fp32_nchw(B, W, H, N, route):
    """The UNet's conv_out: GroupNorm(32) + SiLU over 128 channels -> conv3x3 -> N <= 4 channels as fp32 NCHW (conv_regw.hip's conv_o4_kernel:
    a wave per 32 input channels; RLDM_NO_O4 is read once per process, so the generic kernel is reached through its force-tile switch)."""
    from rangeldm_amd import _lib
    x = _rand(B, 128, W, H, seed=80) * 1.2 - 0.1
    w = _rand(N, 128, 3, 3, seed=81, scale=(128 * 9) ** -0.5)
    b = _rand(N, seed=82, scale=0.1)
    gamma, beta = 1 + 0.2 * _rand(128, seed=83), 0.2 * _rand(128, seed=84)
    _lib.lib().rldm_debug_set_flags2(1 << 26)
    if route == "generic":
        _lib.lib().rldm_debug_set_flags(256 + 2048)
    try:
        y = hip_conv(x, w, b, gamma=gamma, beta=beta, silu=True, eps=1e-5)
    finally:
        _lib.lib().rldm_debug_set_flags2(0)
        _lib.lib().rldm_debug_set_flags(0)

    def ref(q):
        return ops.circ_conv2d(q(ops.group_norm_silu(q(x), gamma, beta, 32, 1e-5)), q(w), b)

    assert rel_l2(y, ref(bf16r)) < 2e-3
    assert rel_l2(y, ref(lambda t: t)) < TOL_F
    assert float((y - ref(bf16r)).abs().max()) < 0.02


@pytest.mark.parametrize("B,W,H", [(4, 512, 64), (2, 128, 16), (3, 64, 32)])
def test_conv_c64_register_weights_statistics(B, W, H, regw_flags):
    """... its (sum, sumsq) side output: one partial per workgroup, accumulated over the run."""
    x = _rand(B, 64, W, H, seed=60)
    w = _rand(64, 64, 3, 3, seed=61, scale=(64 * 9) ** -0.5)
    b = _rand(64, seed=62, scale=0.5)
    y = hip_conv(x, w, b)
    st = hip_conv_stats(x, w, b)
    ref_s = y.double().sum(dim=(2, 3))
    ref_q = (y.double() ** 2).sum(dim=(2, 3))
    assert float((st[..., 0].double() - ref_s).abs().max()) < 1e-4 * W * H
    assert float(((st[..., 1].double() - ref_q).abs() / (ref_q + 1e-6)).max()) < 1e-4


@pytest.mark.parametrize("B,Cin,Cout,W,H,k", [(2, 128, 128, 64, 16, 3), (3, 64, 256, 32, 2, 3), (2, 32, 64, 16, 8, 1),
                                                (1, 128, 128, 256, 16, 3), (2, 256, 256, 64, 4, 3), (16, 128, 256, 32, 2, 3),
                                                (2, 256, 256, 64, 4, 1), (16, 128, 128, 128, 8, 1), (2, 128, 128, 128, 8, 3),
                                                (4, 256, 256, 32, 1, 3), (2, 256, 256, 32, 1, 1), (1, 64, 64, 1024, 64, 3), (16, 5, 128, 256, 16, 3)])
def test_conv_epilogue_statistics(B, Cin, Cout, W, H, k):
    """The per-channel (sum, sumsq) side output that replaces a separate GroupNorm statistics pass: it must equal
    the sums over the bf16 values the conv stored (fixed-order fp32 partial sums -> tight tolerance)."""
    x = _rand(B, Cin, W, H, seed=30)
    w = _rand(Cout, Cin, k, k, seed=31, scale=(Cin * k * k) ** -0.5)
    b = _rand(Cout, seed=32, scale=0.5)
    y = hip_conv(x, w, b)                       # bf16 outputs, widened to fp32
    st = hip_conv_stats(x, w, b)
    ref_s = y.double().sum(dim=(2, 3))
    ref_q = (y.double() ** 2).sum(dim=(2, 3))
    n = W * H
    assert float((st[..., 0].double() - ref_s).abs().max()) < 1e-4 * n
    assert float(((st[..., 1].double() - ref_q).abs() / (ref_q + 1e-6)).max()) < 1e-4


@pytest.mark.parametrize("B,C,N,W,H,gn,res", [(2, 256, 768, 32, 2, True, False),     # L3 attention q/k/v (32-channel tiles)
                                                (16, 128, 384, 128, 8, True, False),   # L1 q/k/v at the bench batch (128-ch tiles)
                                                (4, 256, 768, 64, 4, True, False),     # L2 q/k/v
                                                (2, 256, 256, 32, 2, False, True),     # L3 attention output projection + x
                                                (4, 256, 256, 32, 1, False, True),     # nuScenes L3 output projection (32-pixel tiles)
                                                (16, 128, 128, 128, 8, False, True),   # L1 output projection + x
                                                (3, 512, 256, 16, 4, True, True)])
def test_conv_pointwise_small_route(B, C, N, W, H, gn, res):
    """conv_small.hip, taps == 1: GroupNorm affine folded into the staging, identity residual added in the epilogue."""
    x = _rand(B, C, W, H, seed=40) * 1.7 + 0.4
    w = _rand(N, C, 1, 1, seed=41, scale=C ** -0.5)
    b = _rand(N, seed=42, scale=0.1)
    gamma, beta = (1 + 0.2 * _rand(C, seed=43), 0.2 * _rand(C, seed=44)) if gn else (None, None)
    r = _rand(B, N, W, H, seed=45) if res else None
    y = hip_conv(x, w, b, gamma=gamma, beta=beta, silu=False, eps=1e-6, res=r)

    def ref(q):
        h = q(F.group_norm(q(x), 32, gamma, beta, 1e-6)) if gn else q(x)
        out = ops.circ_conv2d(h, q(w), b, 1, 0)
        return out + q(r) if res else out

    assert rel_l2(y, ref(bf16r)) < TOL_Q
    assert rel_l2(y, ref(lambda t: t)) < TOL_F


def test_conv_gn_no_silu_1x1():
    # attention's group_norm -> Linear path (no SiLU), eps 1e-6
    B, C, W, H = 2, 128, 16, 8
    x = _rand(B, C, W, H, seed=12) * 2 + 1
    w = _rand(C, C, 1, 1, seed=13, scale=C ** -0.5)
    b = _rand(C, seed=14, scale=0.1)
    gamma, beta = 1 + 0.2 * _rand(C, seed=15), 0.2 * _rand(C, seed=16)
    y = hip_conv(x, w, b, gamma=gamma, beta=beta, silu=False, eps=1e-6)
    h = bf16r(F.group_norm(bf16r(x), 32, gamma, beta, 1e-6))
    assert rel_l2(y, ops.circ_conv2d(h, bf16r(w), b, 1, 0)) < TOL_Q


@pytest.mark.parametrize("B,L,C", [(2, 64, 64), (1, 1024, 128), (2, 256, 256), (1, 32, 32), (1, 8, 32), (2, 48, 16),
                                   (1, 4, 64), (1, 100, 8)])
def test_attention_d8(B, L, C):
    qkv = _rand(B, L, 3 * C, seed=20)
    out = hip_attention(qkv, C)
    q, k, v = bf16r(qkv[..., :C] * (1.4426950408889634 / 8 ** 0.5)) / 1.4426950408889634 * 8 ** 0.5, bf16r(qkv[..., C:2 * C]), bf16r(qkv[..., 2 * C:])
    nh = C // 8
    qh, kh, vh = (t.view(B, L, nh, 8).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, L, C)
    assert rel_l2(out, ref) < 1e-2


def test_attention_large_logits():
    # rows with one dominant key (forces the running-max rescale path) and large magnitudes
    B, L, C = 1, 128, 16
    qkv = _rand(B, L, 3 * C, seed=21)
    qkv[:, :, :2 * C] *= 6.0
    qkv[0, 77, C:2 * C] *= 4.0          # a key far outside the first tiles dominates
    out = hip_attention(qkv, C)
    q, k, v = bf16r(qkv[..., :C]), bf16r(qkv[..., C:2 * C]), bf16r(qkv[..., 2 * C:])
    qh, kh, vh = (t.view(B, L, 2, 8).transpose(1, 2) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, L, C)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 3e-2


def _attn_qkv_ref(x, gamma, beta, wqkv, bqkv, eps=1e-5, rounded=True):
    """oracle.unet.attention_block's arithmetic up to (not including) to_out, on token-major x (B, L, C)."""
    B, L, C = x.shape
    r = bf16r if rounded else (lambda t: t)
    y = F.group_norm(r(x).transpose(1, 2), 32, gamma, beta, eps).transpose(1, 2)
    q, k, v = (F.linear(y, r(wqkv[i * C:(i + 1) * C]), bqkv[i * C:(i + 1) * C]) for i in range(3))
    qh, kh, vh = (t.view(B, L, C // 8, 8).transpose(1, 2) for t in (q, k, v))
    return F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, L, C)


# every (L, C) of SURVEY.md A.4 (RangeLDM: 1024 x 128, 256 x 256, 64 x 256; RangeDM: 256 x 512, 64 x 512; nuScenes: 512 x 128,
# 128 x 256, 32 x 256) plus ragged token counts and the small-config widths
@pytest.mark.parametrize("B,L,C", [(2, 1024, 128), (2, 256, 256), (3, 64, 256), (1, 256, 512), (1, 64, 512), (2, 512, 128),
                                   (2, 128, 256), (4, 32, 256), (2, 100, 64), (1, 8, 32), (2, 1000, 128), (1, 48, 16),
                                   # the bench batch: heads of one image share a workgroup there (HG = 2)
                                   (16, 256, 256), (16, 64, 256), (16, 32, 256), (16, 128, 128)])
def test_attention_qkv_fused_kernel(B, L, C):
    """attention_qkv_d8_kernel alone (the kernel the UNet runs; rldm_test_attention above is the unfused fallback)."""
    if C % 32:
        pytest.skip("GroupNorm(32) needs C % 32 == 0")
    x = _rand(B, L, C, seed=30) * 1.5 + 0.3
    gamma, beta = 1 + 0.2 * _rand(C, seed=31), 0.2 * _rand(C, seed=32)
    wqkv = _rand(3 * C, C, seed=33, scale=2.0 * C ** -0.5)          # sharper softmax than unit-variance logits
    bqkv = _rand(3 * C, seed=34, scale=0.1)
    out = hip_attention_qkv(x, gamma, beta, wqkv, bqkv)
    assert torch.isfinite(out).all()
    assert rel_l2(out, _attn_qkv_ref(x, gamma, beta, wqkv, bqkv)) < 1e-2
    assert rel_l2(out, _attn_qkv_ref(x, gamma, beta, wqkv, bqkv, rounded=False)) < TOL_F
    # per head, not just on average: a wrong head would hide in a whole-tensor norm
    ref = _attn_qkv_ref(x, gamma, beta, wqkv, bqkv)
    for h in range(C // 8):
        assert rel_l2(out[..., 8 * h:8 * h + 8], ref[..., 8 * h:8 * h + 8]) < 2e-2, h


def test_attention_qkv_dominant_key():
    """one key far outside the first tiles dominates every query (running-maximum / rescale path of the key loop)."""
    B, L, C = 1, 256, 64
    x = _rand(B, L, C, seed=35)
    x[0, 200] *= 6.0
    gamma, beta = torch.ones(C), torch.zeros(C)
    wqkv = _rand(3 * C, C, seed=36, scale=4.0 * C ** -0.5)
    bqkv = torch.zeros(3 * C)
    out = hip_attention_qkv(x, gamma, beta, wqkv, bqkv)
    assert torch.isfinite(out).all()
    assert rel_l2(out, _attn_qkv_ref(x, gamma, beta, wqkv, bqkv)) < 3e-2


def test_attention_block_matches_reference_crossattention(golden):
    """GroupNorm + the REFERENCE's multi-head CrossAttention(heads=C/8, dim_head=8) + x (tests/golden/mha.npz, produced by
    vae/sgm/modules/attention.py:194-284): fused attention launch + to_out projection + residual on the GPU."""
    from rangeldm_amd.synth import synth_state_dict
    import numpy as np
    g = golden("mha")
    for C, L in ((128, 64), (256, 64), (128, 1024), (256, 1024)):
        tag = f"C{C}_L{L}"
        shapes = {"a.group_norm.weight": (C,), "a.group_norm.bias": (C,), "a.to_out.0.weight": (C, C), "a.to_out.0.bias": (C,)}
        for n in ("to_q", "to_k", "to_v"):
            shapes[f"a.{n}.weight"] = (C, C)
        sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(shapes, prefix=f"mha/{tag}/").items()}
        x = torch.from_numpy(np.asarray(g[f"mha_{tag}_x"], dtype=np.float32))
        ref = torch.from_numpy(np.asarray(g[f"mha_{tag}_y"], dtype=np.float32))
        B, _, W, H = x.shape
        wqkv = torch.cat([sd[f"a.{n}.weight"] * 2.0 for n in ("to_q", "to_k", "to_v")], 0)
        o = hip_attention_qkv(x.view(B, C, W * H).transpose(1, 2).contiguous(), sd["a.group_norm.weight"],
                              sd["a.group_norm.bias"], wqkv, torch.zeros(3 * C))
        o = o.transpose(1, 2).reshape(B, C, W, H)
        y = hip_conv(o, sd["a.to_out.0.weight"].view(C, C, 1, 1), sd["a.to_out.0.bias"], res=x)
        assert rel_l2(y - x, ref - x) < TOL_F, tag          # the attention branch itself, not hidden behind the residual
        assert rel_l2(y, ref) < 5e-3, tag
