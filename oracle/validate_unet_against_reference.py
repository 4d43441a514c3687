#!/usr/bin/env python3
"""Pin the oracle's UNet2DModel wiring, multi-head attention, time embedding and full-length sampling loops against
code that IS in /root/reference, and (re)generate tests/golden/{mha,unetref,traj,presets,inpaint}.npz.

Runs ONLY in the build container (needs /root/reference, read-only); the vectors it writes travel, the reference
does not.  Weights are regenerated from rangeldm_amd.synth on both sides and are never stored.

    python -m oracle.validate_unet_against_reference            # check + write goldens   (~6 min on 8 cores)
    python -m oracle.validate_unet_against_reference --check    # check only
    python -m oracle.validate_unet_against_reference --quick    # skip the 50-step trajectories and RangeDM

What is pinned to what (diffusers itself is absent):
  * `oracle.ops.timestep_embedding(flip_sin_to_cos, freq_shift)`  <->  sgm `get_timestep_embedding`
    (vae/sgm/modules/diffusionmodules/model.py:28-46) == the (False, 1) setting of the same closed form.
  * `oracle.unet.attention_block(head_dim=8)`  <->  GroupNorm + the reference's multi-head
    `CrossAttention(query_dim=C, heads=C//8, dim_head=8)` (vae/sgm/modules/attention.py:194-284: `(h d)` channel
    split, SDPA with scale d**-0.5, `to_out`; q/k/v are bias-free there, so the oracle's q/k/v biases are zero) + x.
  * `oracle.unet.unet_forward` (skip push/pop, layers_per_block+1 up-resnets, temb MLP + per-resnet projection,
    channel bookkeeping of the concatenations, mid block, norm/act/conv_out)  <->  the reference's temb-ful
    skip-concat UNet `Model` (model.py:521-704) after the SAME surgery the reference applies to the diffusers UNet:
    every stride-2 downsampler replaced by `ldm/utils.py::Downsample2D(padding=1, name="op")` (`replace_down`,
    ldm/utils.py:173-203), every conv circular (`Model(circular=True)`), plus -- glue written here, each a few lines --
    GroupNorm eps 1e-6 -> `norm_eps`, and every single-head `AttnBlock` replaced by GroupNorm + `CrossAttention` + x.
    `Model.__init__` as shipped calls `Downsample(block_in, resamp_with_conv, circular=...)` without the required
    `down_single` argument (model.py:589,640 vs :138,113), i.e. it raises for any multi-level net; the constructor is
    run with `functools.partial(Downsample, down_single=False)` / `partial(Upsample, up_single=False)` bound in.
  * 50-step samplers: the reference's own `LDMPipelineRange.__call__` (ldm/pipelines.py:282-383) drives that reference
    `Model`, the sgm `Decoder` and the oracle's scheduler objects (schedulers stay restated: diffusers only).
"""
import argparse
import functools
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import validate_against_reference as V  # noqa: E402
from oracle import ops, unet as o_unet, vae as o_vae, schedulers as o_sched, pipelines as o_pipe  # noqa: E402
from rangeldm_amd.config import UNetConfig, VAEConfig, PRESETS  # noqa: E402
from rangeldm_amd.params import unet_param_shapes, vae_param_shapes  # noqa: E402
from rangeldm_amd.synth import synth_state_dict, normal, uniform  # noqa: E402

GOLD = V.GOLD
check = V.check


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ---- the reference-composed UNet ---------------------------------------------------------------------------------
class RefMultiHeadAttn(nn.Module):
    """GroupNorm -> reference CrossAttention (multi-head, d=8) -> + x.  Only the rearranges and the residual are glue."""

    def __init__(self, attention_mod, sgm, C, head_dim, eps):
        super().__init__()
        self.norm = sgm.Normalize(C)
        self.norm.eps = eps
        self.attn = attention_mod.CrossAttention(query_dim=C, heads=C // head_dim, dim_head=head_dim)

    def forward(self, x):
        B, C, W, H = x.shape
        h = self.norm(x).view(B, C, W * H).transpose(1, 2)
        h = self.attn(h)
        return x + h.transpose(1, 2).reshape(B, C, W, H)


def build_reference_unet(sgm, lu, attention_mod, cfg: UNetConfig):
    """sgm `Model` + the reference's surgery, configured like UNet2DModel(**cfg)."""
    boc = cfg.block_out_channels
    L, ch = len(boc), boc[0]
    assert all(c % ch == 0 for c in boc)
    down_attn = [t == "AttnDownBlock2D" for t in cfg.down_block_types]
    up_attn = [t == "AttnUpBlock2D" for t in cfg.up_block_types]
    assert up_attn == down_attn[::-1], "sgm Model places attention per resolution (both directions)"
    res = cfg.sample_size[0]
    D, U = sgm.Downsample, sgm.Upsample
    sgm.Downsample = functools.partial(D, down_single=False)
    sgm.Upsample = functools.partial(U, up_single=False)
    try:
        m = sgm.Model(ch=ch, out_ch=cfg.out_channels, ch_mult=tuple(c // ch for c in boc),
                      num_res_blocks=cfg.layers_per_block, attn_resolutions=[res >> i for i in range(L) if down_attn[i]],
                      dropout=0.0, in_channels=cfg.in_channels, resolution=res, act="silu", circular=True)
    finally:
        sgm.Downsample, sgm.Upsample = D, U
    assert m.temb_ch == cfg.time_embed_dim
    for mod in m.modules():
        if isinstance(mod, nn.GroupNorm):
            mod.eps = cfg.norm_eps
    for i in range(L - 1):                                       # replace_down, ldm/utils.py:173-203
        m.down[i].downsample = lu.Downsample2D(boc[i], use_conv=True, out_channels=boc[i], padding=1, name="op")

    def mh(C):
        return RefMultiHeadAttn(attention_mod, sgm, C, cfg.attention_head_dim, cfg.norm_eps)

    for i in range(L):
        for lst in (m.down[i].attn, m.up[i].attn):
            for j in range(len(lst)):
                lst[j] = mh(boc[i])
    m.mid.attn_1 = mh(boc[-1]) if cfg.add_attention else nn.Identity()
    return m.eval()


def ref_key(k, L):
    """diffusers UNet2DModel key -> key of the reference-composed Model (inverse of ldm/convert_vae.py-style renames)."""
    k = k.replace("time_embedding.linear_1", "temb.dense.0").replace("time_embedding.linear_2", "temb.dense.1")
    k = k.replace("conv_norm_out", "norm_out").replace("time_emb_proj", "temb_proj").replace("conv_shortcut", "nin_shortcut")
    k = k.replace("mid_block.resnets.0", "mid.block_1").replace("mid_block.resnets.1", "mid.block_2")
    k = k.replace("mid_block.attentions.0", "mid.attn_1")
    p = k.split(".")
    if p[0] in ("down_blocks", "up_blocks"):
        lvl = int(p[1]) if p[0] == "down_blocks" else L - 1 - int(p[1])
        kind = {"resnets": "block", "attentions": "attn", "downsamplers": "downsample", "upsamplers": "upsample"}[p[2]]
        rest = p[4:] if kind in ("downsample", "upsample") else p[3:]
        k = ".".join([p[0].split("_")[0], str(lvl), kind] + rest)
    k = k.replace(".group_norm.", ".norm.")
    for n in ("to_q", "to_k", "to_v", "to_out"):
        k = k.replace(f".{n}.", f".attn.{n}.")
    return k


def load_ref_unet(m, sd, L):
    """Load a diffusers-keyed synthetic state dict; q/k/v biases have no counterpart (CrossAttention is bias-free)."""
    want = dict(m.state_dict())
    got = {}
    for k, v in sd.items():
        if any(k.endswith(f".{n}.bias") for n in ("to_q", "to_k", "to_v")):
            assert not np.any(v), k
            continue
        rk = ref_key(k, L)
        assert rk in want, (k, rk)
        got[rk] = T(v).reshape(want[rk].shape)
    missing = set(want) - set(got)
    assert not missing, sorted(missing)[:5]
    m.load_state_dict(got)


def synth_unet_sd(cfg, prefix):
    sd = synth_state_dict(unet_param_shapes(cfg), prefix=prefix)
    for k in sd:
        if any(k.endswith(f".{n}.bias") for n in ("to_q", "to_k", "to_v")):
            sd[k] = np.zeros_like(sd[k])
    return sd


class RefUNet:
    """unet(x, t).sample over the reference-composed Model (the duck-typed surface of SURVEY.md 8b)."""

    class _Out:
        def __init__(self, s):
            self.sample = s

    def __init__(self, model, cfg, record=None):
        self.m, self.config, self.record = model, cfg, record
        self.dtype, self.device = torch.float32, torch.device("cpu")

    @torch.no_grad()
    def __call__(self, x, t):
        t = torch.as_tensor(t).reshape(-1)
        t = (t * torch.ones(x.shape[0], dtype=t.dtype)) if t.numel() == 1 else t
        if self.record is not None:
            self.record.append(x[:, :self.config.out_channels].clone())
        return self._Out(self.m(x, t))


class RefVAE:
    """vae.decode / vae.encode over sgm Decoder / Encoder + distributions.py."""

    class _Dec:
        def __init__(self, s):
            self.sample = s

    class _Enc:
        def __init__(self, d):
            self.latent_dist = d

    def __init__(self, enc, dec, dist_mod, cfg, record=None):
        self.enc, self.dec, self.dist, self.config, self.record = enc, dec, dist_mod, cfg, record

    @torch.no_grad()
    def decode(self, z):
        if self.record is not None:
            self.record.append(z.clone())
        return self._Dec(self.dec(z))

    @torch.no_grad()
    def encode(self, x):
        return self._Enc(self.dist.DiagonalGaussianDistribution(self.enc(x)))


def build_reference_vae(sgm, vcfg, vsd):
    kw = dict(attn_type="none", double_z=True, z_channels=vcfg.z_channels, resolution=256, in_channels=vcfg.in_channels,
              out_ch=vcfg.out_channels, ch=vcfg.ch, ch_mult=list(vcfg.ch_mult), num_res_blocks=vcfg.num_res_blocks,
              attn_resolutions=[], dropout=0.0, act="silu", circular=True)
    enc, dec = sgm.Encoder(**kw), sgm.Decoder(**kw)
    enc.load_state_dict(V.diffusers_sd_to_sgm(vsd, enc.state_dict(), "encoder"))
    dec.load_state_dict(V.diffusers_sd_to_sgm(vsd, dec.state_dict(), "decoder"))
    return enc.eval(), dec.eval()


def sgm_sinusoid(cfg_kw):
    """UNetConfig with the time-embedding sinusoid of the reference's own get_timestep_embedding."""
    return UNetConfig(**cfg_kw, flip_sin_to_cos=False, freq_shift=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    gold = {}
    sgm = V.import_sgm_model()
    lu = V.import_ldm_utils()
    att = sys.modules["sgm.modules.attention"]
    dist = V._load("ref_distributions", os.path.join(V.REF, "vae", "sgm", "modules", "distributions", "distributions.py"))

    print("== sinusoidal timestep embedding vs sgm get_timestep_embedding (model.py:28-46)")
    for dim in (32, 128):
        t = torch.tensor([0, 1, 20, 480, 980, 999])
        check(f"get_timestep_embedding dim {dim} == Timesteps(flip_sin_to_cos=False, freq_shift=1)",
              ops.timestep_embedding(t, dim, flip_sin_to_cos=False, freq_shift=1), sgm.get_timestep_embedding(t, dim), 1e-4)
        # (not 0: the two spell the exponent -ln(1e4)*i/(half-1) in a different fp32 operation order; at t = 999 the
        #  argument's ulp is 6e-5)
    e0 = ops.timestep_embedding(torch.tensor([480]), 128)
    e1 = ops.timestep_embedding(torch.tensor([480]), 128, flip_sin_to_cos=False, freq_shift=0)
    check("flip_sin_to_cos swaps the halves", torch.cat([e0[:, 64:], e0[:, :64]], 1), e1, 0.0)

    print("== multi-head attention (d=8) vs GroupNorm + sgm CrossAttention(heads=C/8, dim_head=8) + x")
    for C, (W, H) in ((128, (16, 4)), (256, (16, 4)), (128, (64, 16)), (256, (64, 16))):
        tag = f"C{C}_L{W * H}"
        blk = RefMultiHeadAttn(att, sgm, C, 8, 1e-5)
        shapes = {"a.group_norm.weight": (C,), "a.group_norm.bias": (C,), "a.to_out.0.weight": (C, C), "a.to_out.0.bias": (C,)}
        for n in ("to_q", "to_k", "to_v"):
            shapes[f"a.{n}.weight"] = (C, C)
        sd = synth_state_dict(shapes, prefix=f"mha/{tag}/")
        for n in ("to_q", "to_k", "to_v"):
            sd[f"a.{n}.weight"] *= 2.0                      # sharper softmax than unit-variance logits
            sd[f"a.{n}.bias"] = np.zeros(C, np.float32)
        with torch.no_grad():
            blk.norm.weight.copy_(T(sd["a.group_norm.weight"]))
            blk.norm.bias.copy_(T(sd["a.group_norm.bias"]))
            for n in ("to_q", "to_k", "to_v"):
                getattr(blk.attn, n).weight.copy_(T(sd[f"a.{n}.weight"]))
            blk.attn.to_out[0].weight.copy_(T(sd["a.to_out.0.weight"]))
            blk.attn.to_out[0].bias.copy_(T(sd["a.to_out.0.bias"]))
        x = T(normal(21, f"mha/{tag}/x", (2 if W * H == 64 else 1, C, W, H))) * 1.5 + 0.3
        with torch.no_grad():
            ref = blk(x)
        mine = o_unet.attention_block({k: T(v) for k, v in sd.items()}, "a", x, 32, 1e-5, 8)
        check(f"CrossAttention C={C} L={W * H}", mine, ref, 3e-5)
        gold[f"mha_{tag}_x"] = x.numpy().astype(np.float16 if W * H == 1024 else np.float32)
        gold[f"mha_{tag}_y"] = ref.numpy()
        if W * H == 1024:                                   # the stored input is the fp16-rounded one: recompute on it
            xr = T(gold[f"mha_{tag}_x"]).float()
            with torch.no_grad():
                gold[f"mha_{tag}_y"] = blk(xr).numpy().astype(np.float16)

    print("== UNet wiring vs the reference-composed sgm Model (skip concat, temb, up/down, multi-head attention)")
    small_kw = dict(sample_size=(64, 8), block_out_channels=(32, 32, 64, 64))
    dm_kw = dict(sample_size=(128, 32), in_channels=3, out_channels=2, block_out_channels=(32, 32, 64, 64, 96, 96),
                 down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
                 up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4)
    for name, kw, B in (("small", small_kw, 2), ("rangedm_topology", dm_kw, 1)):
        for sinus in ("sgm", "unet2d"):
            cfg = sgm_sinusoid(kw) if sinus == "sgm" else UNetConfig(**kw)
            sd = synth_unet_sd(cfg, f"ref/{name}.")
            m = build_reference_unet(sgm, lu, att, cfg)
            load_ref_unet(m, sd, len(cfg.block_out_channels))
            x = T(normal(31, f"ref/{name}/x", (B, cfg.in_channels, *cfg.sample_size)))
            ts = torch.tensor([480] * B) if B == 1 else torch.tensor([480, 37])
            if sinus == "unet2d":                          # the one difference to UNet2DModel, patched on the reference side
                orig = sgm.get_timestep_embedding
                sgm.get_timestep_embedding = lambda t, d: ops.timestep_embedding(t, d)
            try:
                with torch.no_grad():
                    ref = m(x, ts)
            finally:
                if sinus == "unet2d":
                    sgm.get_timestep_embedding = orig
            mine = o_unet.unet_forward({k: T(v) for k, v in sd.items()}, cfg, x, ts)
            check(f"Model({name}) sinusoid={sinus}", mine, ref, 2e-5 * float(ref.abs().max()))
            gold[f"unetref_{name}_{sinus}_x"], gold[f"unetref_{name}_{sinus}_t"] = x.numpy(), ts.numpy()
            gold[f"unetref_{name}_{sinus}_eps"] = ref.numpy()

    print("== full-width RangeLDM UNet (30.1 M parameters) vs the reference-composed Model")
    full = UNetConfig()
    full_sgm = sgm_sinusoid({})
    fsd = synth_unet_sd(full, "ref/full.")
    fm = build_reference_unet(sgm, lu, att, full)
    load_ref_unet(fm, fsd, 4)
    n_ref = sum(p.numel() for p in fm.parameters())
    assert n_ref == 30135684 - 3 * (5 * 128 + 11 * 256), n_ref      # minus the q/k/v biases of the 16 attention blocks
    print(f"  [ok] reference-composed Model has {n_ref} parameters = 30 135 684 - q/k/v biases")
    fsd_t = {k: T(v) for k, v in fsd.items()}
    xu = T(normal(9, "unet/x", (1, 5, 256, 16)))
    with torch.no_grad():
        ref = fm(xu, torch.tensor([480]))
    check("Model(RangeLDM full width) sinusoid=sgm", o_unet.unet_forward(fsd_t, full_sgm, xu, 480), ref,
          2e-5 * float(ref.abs().max()))
    gold["unetref_full_x"], gold["unetref_full_t"], gold["unetref_full_sgm_eps"] = xu.numpy(), np.array([480]), ref.numpy()

    print("== the other presets at full width (SURVEY.md A.1)")
    # upsample: 12 input channels (4 latent + 8 folded condition)
    ucfg = sgm_sinusoid(dict(in_channels=12))
    usd = synth_unet_sd(ucfg, "ref/up.")
    um = build_reference_unet(sgm, lu, att, ucfg)
    load_ref_unet(um, usd, 4)
    xup = T(normal(9, "up/x", (1, 12, 256, 16)))
    with torch.no_grad():
        ref = um(xup, torch.tensor([700]))
    check("Model(upsample full width, 12 ch)", o_unet.unet_forward({k: T(v) for k, v in usd.items()}, ucfg, xup, 700), ref,
          2e-5 * float(ref.abs().max()))
    gold["presets_up_x"], gold["presets_up_eps"] = xup.numpy(), ref.numpy()
    # nuScenes-shape VAE decode vs sgm Decoder
    vcfg = VAEConfig()
    vsd = synth_state_dict(vae_param_shapes(vcfg), prefix="vae.")
    enc, dec = build_reference_vae(sgm, vcfg, vsd)
    zn = T(normal(6, "vae/znusc", (1, 4, 256, 8)))
    with torch.no_grad():
        ref = dec(zn)
    check("sgm.Decoder (1,4,256,8) nuScenes", o_vae.vae_decode(V.vsd_t(vsd), vcfg, zn), ref, 5e-5)
    gold["presets_nusc_z"], gold["presets_nusc_image_f16"] = zn.numpy(), ref.numpy().astype(np.float16)
    if not args.quick:
        rcfg_kw = {k: v for k, v in PRESETS["RangeDM"]["unet"].to_dict().items()
                   if k not in ("flip_sin_to_cos", "freq_shift")}
        rcfg = sgm_sinusoid(rcfg_kw)
        rsd = synth_unet_sd(rcfg, "ref/rangedm.")
        rm = build_reference_unet(sgm, lu, att, rcfg)
        load_ref_unet(rm, rsd, 6)
        xr = T(normal(9, "rangedm/x", (1, 3, 1024, 64)).astype(np.float16).astype(np.float32))
        t0 = time.time()
        with torch.no_grad():
            ref = rm(xr, torch.tensor([900]))
        print(f"  (reference RangeDM forward {time.time() - t0:.1f} s)")
        check("Model(RangeDM full size 1024x64, 113.7 M)", o_unet.unet_forward({k: T(v) for k, v in rsd.items()}, rcfg, xr, 900),
              ref, 2e-5 * float(ref.abs().max()))
        gold["presets_rangedm_x_f16"], gold["presets_rangedm_eps"] = xr.numpy().astype(np.float16), ref.numpy()
        del rm, rsd

    print("== inpainting: LDMUpscalePipelineRange.encode_masked_image + loop (ldm/pipelines.py:406-412,466-507)")
    lp = V.import_ldm_pipelines()
    icfg_small = UNetConfig(sample_size=(32, 8), in_channels=9, out_channels=4, block_out_channels=(32, 32, 64, 64))
    isd = synth_state_dict(unet_param_shapes(icfg_small), prefix="smallinp.")
    ounet = o_unet.OracleUNet(icfg_small, isd)
    rvae = RefVAE(enc, dec, dist, vcfg)
    pipe = lp.LDMUpscalePipelineRange(vae=rvae, unet=ounet, scheduler=o_sched.OracleDDPMScheduler())
    img = T(normal(41, "inp/image", (2, 2, 128, 32)))
    mask = (T(uniform(41, "inp/mask", (2, 1, 128, 32))) > 0.2).float()
    torch.manual_seed(77)
    cond_ref = pipe.encode_masked_image(img * mask, mask)
    torch.manual_seed(77)
    enc_noise = torch.randn(2, 4, 32, 8)
    ovae = o_vae.OracleVAE(vcfg, vsd)
    lat = ovae.encode(img * mask).latent_dist.sample(noise=enc_noise) * vcfg.scaling_factor
    mine = torch.cat([lat, torch.nn.functional.interpolate(mask, size=lat.shape[-2:])], 1)
    check("encode_masked_image", mine, cond_ref, 5e-5)
    g = torch.Generator().manual_seed(43)
    torch.manual_seed(44)                                  # encode noise first, then the DDPM step noise (global RNG)
    ref = pipe(image=img * mask, mask=mask, batch_size=2, generator=g, num_inference_steps=3, output_type="torch")
    g = torch.Generator().manual_seed(43)
    x_T = torch.randn((2, 4, 32, 8), generator=g)
    torch.manual_seed(44)
    enc_noise2 = torch.randn(2, 4, 32, 8)
    zs = [torch.randn(2, 4, 32, 8) for _ in range(2)] + [None]
    lat = ovae.encode(img * mask).latent_dist.sample(noise=enc_noise2) * vcfg.scaling_factor
    cond = torch.cat([lat, torch.nn.functional.interpolate(mask, size=lat.shape[-2:])], 1)
    mine = o_pipe.ldm_pipeline(ovae, ounet, o_sched.OracleDDPMScheduler(), x_T, 3, pos_encoding=False, step_noise=zs, cond=cond)
    check("LDMUpscalePipelineRange (mask path, DDPM, 3 steps)", mine, ref, 1e-4)
    gold["inpaint_image"], gold["inpaint_mask"] = img.numpy(), mask.numpy()
    gold["inpaint_enc_noise"], gold["inpaint_cond_ref"] = enc_noise.numpy(), cond_ref.numpy()
    gold["inpaint_x_T"], gold["inpaint_enc_noise2"] = x_T.numpy(), enc_noise2.numpy()
    gold["inpaint_step_noise"], gold["inpaint_image_ref"] = torch.stack(zs[:2]).numpy(), ref.numpy()

    if not args.quick:
        print("== 50-step full-width RangeLDM samplers: reference LDMPipelineRange loop + reference Model + sgm Decoder")
        for sched_name in ("ddim", "ddpm"):
            xs, zrec = [], []
            runet = RefUNet(fm, full_sgm, record=xs)
            rv = RefVAE(enc, dec, dist, vcfg, record=zrec)
            x_T = T(normal(51, f"traj/{sched_name}/x_T", (1, 4, 256, 16)))
            step_z = [T(normal(52, f"traj/z/{i}", (1, 4, 256, 16))) for i in range(50)]
            lp.randn_tensor = lambda shape, generator=None, device=None, dtype=None, layout=None: x_T.clone()
            if sched_name == "ddim":
                sched = o_sched.OracleDDIMScheduler()
            else:                                          # inject z_i instead of drawing from the global RNG
                class InjectedDDPM(o_sched.OracleDDPMScheduler):
                    def step(self, eps, t, x, **kw):
                        i = int((self.timesteps == int(t)).nonzero()[0])
                        return super().step(eps, t, x, noise=step_z[i] if int(t) > 0 else None)
                sched = InjectedDDPM()
            pipe = lp.LDMPipelineRange(vae=rv, unet=runet, scheduler=sched, pos_encoding=True)
            t0 = time.time()
            ref_img = pipe(batch_size=1, generator=None, num_inference_steps=50, output_type="torch")
            print(f"  ({sched_name}: reference loop {time.time() - t0:.0f} s, |x_0| max {float(zrec[-1].abs().max()):.1f},"
                  f" image |max| {float(ref_img.abs().max()):.2f})")
            ofull = o_unet.OracleUNet(full_sgm, fsd)
            osched = o_sched.OracleDDIMScheduler() if sched_name == "ddim" else o_sched.OracleDDPMScheduler()
            zs = None if sched_name == "ddim" else step_z[:49] + [None]
            mine = o_pipe.ldm_pipeline(ovae, ofull, osched, x_T, 50, pos_encoding=True, step_noise=zs)
            check(f"LDMPipelineRange 50 steps ({sched_name}) image", mine, ref_img, 2e-3 * float(ref_img.abs().max()))
            gold[f"traj_{sched_name}_latent_ref"] = (zrec[-1] * vcfg.scaling_factor).numpy()     # x_0 before the /0.18215
            gold[f"traj_{sched_name}_image_ref_f16"] = ref_img.numpy().astype(np.float16)
            for i in (1, 10, 25, 40, 49):
                gold[f"traj_{sched_name}_x_step{i}"] = xs[i].numpy()                               # UNet input latent of step i

    bad = [c for c in V.CHECKS if not c[3]]
    print(f"\n{len(V.CHECKS) - len(bad)}/{len(V.CHECKS)} checks passed")
    if bad:
        sys.exit(1)
    if not args.check:
        groups = {}
        for k, v in gold.items():
            groups.setdefault(k.split("_", 1)[0], {})[k] = np.ascontiguousarray(v)
        for gname, d in groups.items():
            if args.quick and gname in ("traj", "presets"):
                continue
            path = os.path.join(GOLD, f"{gname}.npz")
            np.savez_compressed(path, **d)
            print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
