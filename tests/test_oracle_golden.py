"""CPU: the oracle reproduces the golden vectors that oracle/validate_against_reference.py captured FROM THE
REFERENCE'S OWN MODULES (sgm Encoder/Decoder/ResnetBlock/AttnBlock/Downsample/Upsample, ldm/utils.py, ldm/encoders.py,
ldm/pipelines.py loops) -- this is what pins the oracle on machines where /root/reference does not exist."""
import numpy as np
import pytest
import torch

from rangeldm_amd.config import UNetConfig, VAEConfig
from rangeldm_amd.params import unet_param_shapes, vae_param_shapes, count_params, sgm_to_diffusers_vae_key
from rangeldm_amd.synth import synth_state_dict
from oracle import ops, unet as o_unet, vae as o_vae, schedulers as o_sched, pipelines as o_pipe


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_param_counts_match_reference_checkpoints():
    # README.md:8 -- RangeLDM checkpoint 115 MB == 30 135 684 fp32 params; VAE sizes from SURVEY.md 8c probe
    assert count_params(unet_param_shapes(UNetConfig())) == 30135684
    assert count_params(unet_param_shapes(UNetConfig(in_channels=12))) == 30143748
    v = vae_param_shapes(VAEConfig())
    assert count_params({k: s for k, s in v.items() if k.startswith("encoder.")}) == 5341320
    assert count_params({k: s for k, s in v.items() if k.startswith("decoder.")}) == 7989570


def test_sgm_key_map():
    m = sgm_to_diffusers_vae_key
    assert m("encoder.down.1.block.0.nin_shortcut.weight") == "encoder.down_blocks.1.resnets.0.conv_shortcut.weight"
    assert m("encoder.down.0.downsample.conv.bias") == "encoder.down_blocks.0.downsamplers.0.conv.bias"
    assert m("decoder.up.2.block.1.norm1.weight") == "decoder.up_blocks.0.resnets.1.norm1.weight"
    assert m("decoder.up.1.upsample.conv.weight") == "decoder.up_blocks.1.upsamplers.0.conv.weight"
    assert m("decoder.mid.block_2.conv1.weight") == "decoder.mid_block.resnets.1.conv1.weight"
    assert m("decoder.norm_out.bias") == "decoder.conv_norm_out.bias"
    assert m("loss.discriminator.main.0.weight") is None


def test_leaf_ops(golden):
    g = golden("leaf")
    x = T(g["leaf_x"])
    assert torch.equal(ops.downsample_unet(x, T(g["leaf_down_unet_w"]), T(g["leaf_down_unet_b"])), T(g["leaf_down_unet_y"]))
    assert torch.equal(ops.downsample_vae(x, T(g["leaf_down_vae_w"]), T(g["leaf_down_vae_b"])), T(g["leaf_down_vae_y"]))
    assert torch.equal(ops.upsample_conv(x, T(g["leaf_up_w"]), T(g["leaf_up_b"])), T(g["leaf_up_y"]))


def test_circular_conv_is_wrap_w_zero_h():
    # independent restatement by explicit index arithmetic (no F.pad) on a tiny case
    x = torch.randn(1, 2, 5, 3)
    w = torch.randn(3, 2, 3, 3)
    y = ops.circ_conv2d(x, w, None)
    ref = torch.zeros(1, 3, 5, 3)
    for o in range(3):
        for wi in range(5):
            for hi in range(3):
                acc = 0.0
                for c in range(2):
                    for i in range(3):
                        for j in range(3):
                            hh = hi + j - 1
                            if 0 <= hh < 3:
                                acc += float(x[0, c, (wi + i - 1) % 5, hh]) * float(w[o, c, i, j])
                ref[0, o, wi, hi] = acc
    assert torch.allclose(y, ref, atol=1e-5)


def test_resnet_block(golden):
    g = golden("resnet")
    sd = {k[len("resnet_"):]: T(v) for k, v in g.items() if k.startswith("resnet_r.")}
    y = o_unet.resnet_block(sd, "r", T(g["resnet_x"]), T(g["resnet_temb"]), 32, 1e-5)
    assert (y - T(g["resnet_y"])).abs().max() < 5e-6


def test_attention_block_single_head_and_multihead(golden):
    g = golden("attn")
    sd = {k[len("attn_"):]: T(v) for k, v in g.items() if k.startswith("attn_a.")}
    x = T(g["attn_x"])
    y = o_unet.attention_block(sd, "a", x, 32, 1e-6, 64)
    assert (y - T(g["attn_y_single_head"])).abs().max() < 1e-5
    # multi-head (d=8) == the single-head formula applied per 8-channel slice
    y8 = o_unet.attention_block(sd, "a", x, 32, 1e-6, 8)
    import torch.nn.functional as F
    B, C, W, H = x.shape
    n = F.group_norm(x, 32, sd["a.group_norm.weight"], sd["a.group_norm.bias"], 1e-6).view(B, C, -1).transpose(1, 2)
    q, k, v = (F.linear(n, sd[f"a.{m}.weight"], sd[f"a.{m}.bias"]) for m in ("to_q", "to_k", "to_v"))
    outs = []
    for h in range(C // 8):
        s = slice(8 * h, 8 * h + 8)
        p = torch.softmax(q[..., s] @ k[..., s].transpose(1, 2) / 8 ** 0.5, dim=-1)
        outs.append(p @ v[..., s])
    o = F.linear(torch.cat(outs, -1), sd["a.to_out.0.weight"], sd["a.to_out.0.bias"])
    ref = o.transpose(1, 2).reshape(B, C, W, H) + x
    assert (y8 - ref).abs().max() < 1e-5


def test_vae_matches_sgm_goldens(golden):
    g = golden("vae")
    cfg = VAEConfig()
    sd = {k: T(v) for k, v in synth_state_dict(vae_param_shapes(cfg), prefix="vae.").items()}
    assert (o_vae.vae_encode(sd, cfg, T(g["vae_x"])) - T(g["vae_moments_ref"])).abs().max() < 2e-5
    assert (o_vae.vae_decode(sd, cfg, T(g["vae_z"])) - T(g["vae_image_ref"])).abs().max() < 2e-5


def test_diag_gaussian_and_condition_encoder(golden):
    g, v = golden("dg"), golden("vae")
    s = o_vae.DiagonalGaussian(T(v["vae_moments_ref"])).sample(noise=T(g["dg_noise"]))
    assert torch.equal(s, T(g["dg_sample_ref"]))
    c = golden("cond")
    assert torch.equal(o_pipe.sparse_range_image_encoder2(T(c["cond_x"])).contiguous(), T(c["cond_y_ref"]))
    # closed form: out[b, (w%4)*C + c, w//4, h] = in[b, c, w, h]
    x = T(c["cond_x"])
    y = T(c["cond_y_ref"])
    assert y[1, (7 % 4) * 2 + 1, 7 // 4, 3] == x[1, 1, 7, 3]


def test_scheduler_known_answers():
    # SURVEY.md B.4 (computed from the closed form of diffusers' schedulers)
    s = o_sched.OracleDDIMScheduler()
    s.set_timesteps(50)
    assert s.timesteps[:3].tolist() == [980, 960, 940] and s.timesteps[-1] == 0
    ac = s.alphas_cumprod
    for t, v in ((0, 0.9998999834), (20, 0.9937353134), (500, 0.07779665291), (980, 5.903752026e-05), (999, 4.035830352e-05)):
        assert abs(float(ac[t]) - v) < 1e-7 * max(1, v / 1e-5)
    x = torch.tensor([1.5409961, -0.2934289, -2.1787894, 0.5684313]).view(1, 1, 2, 2)
    e = torch.tensor([-1.0845224, -1.3985955, 0.4033468, 0.8380263]).view(1, 1, 2, 2)
    out = s.step(e, 980, x).prev_sample.flatten()
    assert torch.allclose(out, torch.tensor([2.1102533, -0.0538026, -2.7386351, 0.5099728]), atol=2e-6)
    p = o_sched.OracleDDPMScheduler()
    p.set_timesteps(50)
    mu = p.step(e, 980, x, noise=torch.zeros_like(x)).prev_sample.flatten()
    assert torch.allclose(mu, torch.tensor([2.3034992, 0.1954067, -2.8105054, 0.3606487]), atol=2e-6)
    assert abs(p.coefficients(980)[2] - 0.5697414) < 1e-6
    s.set_timesteps(10)
    assert s.timesteps.tolist() == [900, 800, 700, 600, 500, 400, 300, 200, 100, 0]
    # add_noise (ldm/train_unconditional.py:498)
    t = torch.tensor([0, 999])
    xn = p.add_noise(torch.ones(2, 1, 1, 1), torch.full((2, 1, 1, 1), 2.0), t).flatten()
    assert torch.allclose(xn, ac[t] ** 0.5 + 2 * (1 - ac[t]) ** 0.5)


def _small(in_ch, out_ch, prefix):
    cfg = UNetConfig(sample_size=(32, 8), in_channels=in_ch, out_channels=out_ch, block_out_channels=(32, 32, 64, 64))
    return o_unet.OracleUNet(cfg, synth_state_dict(unet_param_shapes(cfg), prefix=prefix))


def test_pipelines_match_reference_loops(golden):
    vcfg = VAEConfig()
    vae = o_vae.OracleVAE(vcfg, synth_state_dict(vae_param_shapes(vcfg), prefix="vae."))
    g = golden("ldm")
    zs = [z for z in T(g["ldm_step_noise"])] + [None]
    img = o_pipe.ldm_pipeline(vae, _small(5, 4, "small."), o_sched.OracleDDPMScheduler(), T(g["ldm_x_T"]), 4,
                              pos_encoding=True, step_noise=zs)
    assert (img - T(g["ldm_image_ref"])).abs().max() < 1e-4
    g = golden("ddim")
    img = o_pipe.ddim_pipeline(_small(3, 2, "smalldm."), o_sched.OracleDDIMScheduler(), T(g["ddim_x_T"]), 5)
    assert (img - T(g["ddim_image_ref"])).abs().max() < 1e-4
    g = golden("ddpmpix")
    zs = [z for z in T(g["ddpmpix_step_noise"])] + [None]
    img = o_pipe.ddpm_pipeline(_small(3, 3, "smallpx."), o_sched.OracleDDPMScheduler(), T(g["ddpmpix_x_T"]), 4,
                               step_noise=zs)
    assert (img - T(g["ddpmpix_image_ref"])).abs().max() < 1e-4
    drawn = o_pipe.ddpm_pipeline(_small(3, 3, "smallpx."), o_sched.OracleDDPMScheduler(),
                                 torch.randn((2, 3, 32, 8), generator=(gen := torch.Generator().manual_seed(41))), 4,
                                 generator=gen)
    assert (drawn - T(g["ddpmpix_image_ref"])).abs().max() < 1e-4
    g = golden("up")
    zs = [z for z in T(g["up_step_noise"])] + [None]
    img = o_pipe.ldm_pipeline(vae, _small(12, 4, "smallup."), o_sched.OracleDDPMScheduler(), T(g["up_x_T"]), 3,
                              pos_encoding=False, step_noise=zs,
                              cond=o_pipe.sparse_range_image_encoder2(T(g["up_cond"])))
    assert (img - T(g["up_image_ref"])).abs().max() < 1e-4


def test_unet_full_shape_regression(golden):
    g = golden("unet")
    cfg = UNetConfig()
    sd = {k: T(v) for k, v in synth_state_dict(unet_param_shapes(cfg)).items()}
    eps = o_unet.unet_forward(sd, cfg, T(g["unet_x"]), int(g["unet_t"][0]))
    assert eps.shape == (1, 4, 256, 16)
    assert (eps - T(g["unet_eps_oracle"])).abs().max() < 1e-4


def test_timestep_embedding_closed_form():
    import math
    e = ops.timestep_embedding(torch.tensor([980]), 128)
    assert e.shape == (1, 128)
    for i in (0, 1, 63):
        f = math.exp(-math.log(10000.0) * i / 64)
        assert abs(float(e[0, i]) - math.cos(980 * f)) < 1e-4 and abs(float(e[0, 64 + i]) - math.sin(980 * f)) < 1e-4


# ---- vectors produced by oracle/validate_unet_against_reference.py from code that IS in the reference: the multi-head
# CrossAttention (vae/sgm/modules/attention.py:194-284), get_timestep_embedding and the skip-concat / temb UNet `Model`
# (vae/sgm/modules/diffusionmodules/model.py:28-46, 521-704) after the reference's own surgery, and the reference's
# LDMPipelineRange / LDMUpscalePipelineRange loops driving them for 50 steps -----------------------------------------
def _zero_qkv_bias(sd):
    for k in sd:
        if any(k.endswith(f".{n}.bias") for n in ("to_q", "to_k", "to_v")):
            sd[k] = np.zeros_like(sd[k])
    return sd


def ref_unet_sd(cfg, prefix):
    """the weights of the reference-composed Model: synthetic, q/k/v biases zero (CrossAttention has none)."""
    return _zero_qkv_bias(synth_state_dict(unet_param_shapes(cfg), prefix=prefix))


SGM_SINUSOID = dict(flip_sin_to_cos=False, freq_shift=1)      # get_timestep_embedding of the reference (model.py:28-46)
REF_UNETS = {
    "small": dict(sample_size=(64, 8), block_out_channels=(32, 32, 64, 64)),
    "rangedm_topology": dict(sample_size=(128, 32), in_channels=3, out_channels=2, block_out_channels=(32, 32, 64, 64, 96, 96),
                             down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
                             up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4),
}


def test_multihead_attention_matches_reference_crossattention(golden):
    g = golden("mha")
    for C, L, tol in ((128, 64, 3e-5), (256, 64, 3e-5), (128, 1024, 3e-3), (256, 1024, 3e-3)):     # L=1024 stored as fp16
        tag = f"C{C}_L{L}"
        shapes = {"a.group_norm.weight": (C,), "a.group_norm.bias": (C,), "a.to_out.0.weight": (C, C), "a.to_out.0.bias": (C,)}
        for n in ("to_q", "to_k", "to_v"):
            shapes[f"a.{n}.weight"] = (C, C)
        sd = synth_state_dict(shapes, prefix=f"mha/{tag}/")
        for n in ("to_q", "to_k", "to_v"):
            sd[f"a.{n}.weight"] = sd[f"a.{n}.weight"] * 2.0
            sd[f"a.{n}.bias"] = np.zeros(C, np.float32)
        y = o_unet.attention_block({k: T(v) for k, v in sd.items()}, "a", T(g[f"mha_{tag}_x"]).float(), 32, 1e-5, 8)
        assert (y - T(g[f"mha_{tag}_y"]).float()).abs().max() < tol, tag


@pytest.mark.parametrize("name", list(REF_UNETS))
@pytest.mark.parametrize("sinus", ["sgm", "unet2d"])
def test_unet_wiring_matches_reference_model(golden, name, sinus):
    g = golden("unetref")
    cfg = UNetConfig(**REF_UNETS[name], **(SGM_SINUSOID if sinus == "sgm" else {}))
    sd = {k: T(v) for k, v in ref_unet_sd(cfg, f"ref/{name}.").items()}
    eps = o_unet.unet_forward(sd, cfg, T(g[f"unetref_{name}_{sinus}_x"]), T(g[f"unetref_{name}_{sinus}_t"]))
    ref = T(g[f"unetref_{name}_{sinus}_eps"])
    assert (eps - ref).abs().max() < 2e-5 * float(ref.abs().max())


def test_unet_full_width_matches_reference_model(golden):
    g = golden("unetref")
    cfg = UNetConfig(**SGM_SINUSOID)
    sd = {k: T(v) for k, v in ref_unet_sd(cfg, "ref/full.").items()}
    eps = o_unet.unet_forward(sd, cfg, T(g["unetref_full_x"]), int(g["unetref_full_t"][0]))
    ref = T(g["unetref_full_sgm_eps"])
    assert (eps - ref).abs().max() < 2e-5 * float(ref.abs().max())


def test_other_presets_match_reference_modules(golden):
    from rangeldm_amd.config import PRESETS
    g = golden("presets")
    cfg = UNetConfig(in_channels=12, **SGM_SINUSOID)
    sd = {k: T(v) for k, v in ref_unet_sd(cfg, "ref/up.").items()}
    ref = T(g["presets_up_eps"])
    assert (o_unet.unet_forward(sd, cfg, T(g["presets_up_x"]), 700) - ref).abs().max() < 2e-5 * float(ref.abs().max())
    vcfg = VAEConfig()
    vsd = {k: T(v) for k, v in synth_state_dict(vae_param_shapes(vcfg), prefix="vae.").items()}
    img = o_vae.vae_decode(vsd, vcfg, T(g["presets_nusc_z"]))
    assert (img - T(g["presets_nusc_image_f16"]).float()).abs().max() < 3e-3
    kw = {k: v for k, v in PRESETS["RangeDM"]["unet"].to_dict().items() if k not in SGM_SINUSOID}
    rcfg = UNetConfig(**kw, **SGM_SINUSOID)
    rsd = {k: T(v) for k, v in ref_unet_sd(rcfg, "ref/rangedm.").items()}
    ref = T(g["presets_rangedm_eps"])
    eps = o_unet.unet_forward(rsd, rcfg, T(g["presets_rangedm_x_f16"]).float(), 900)
    assert (eps - ref).abs().max() < 2e-5 * float(ref.abs().max())


def test_inpainting_mask_path_matches_reference(golden):
    """LDMUpscalePipelineRange.encode_masked_image + loop (ldm/pipelines.py:406-412, 466-507) on sgm Encoder/Decoder."""
    import torch.nn.functional as F
    g = golden("inpaint")
    vcfg = VAEConfig()
    vae = o_vae.OracleVAE(vcfg, synth_state_dict(vae_param_shapes(vcfg), prefix="vae."))
    img, mask = T(g["inpaint_image"]), T(g["inpaint_mask"])
    lat = vae.encode(img * mask).latent_dist.sample(noise=T(g["inpaint_enc_noise"])) * vcfg.scaling_factor
    cond = torch.cat([lat, F.interpolate(mask, size=lat.shape[-2:])], 1)
    assert (cond - T(g["inpaint_cond_ref"])).abs().max() < 5e-5
    lat = vae.encode(img * mask).latent_dist.sample(noise=T(g["inpaint_enc_noise2"])) * vcfg.scaling_factor
    cond = torch.cat([lat, F.interpolate(mask, size=lat.shape[-2:])], 1)
    zs = [z for z in T(g["inpaint_step_noise"])] + [None]
    out = o_pipe.ldm_pipeline(vae, _small(9, 4, "smallinp."), o_sched.OracleDDPMScheduler(), T(g["inpaint_x_T"]), 3,
                              pos_encoding=False, step_noise=zs, cond=cond)
    assert (out - T(g["inpaint_image_ref"])).abs().max() < 1e-4


@pytest.mark.parametrize("sched", ["ddim", "ddpm"])
def test_50_step_full_width_sampler_matches_reference_loop(golden, sched):
    """The headline workload at batch 1: 50 steps of the full-width RangeLDM UNet + VAE decode.  The golden is the
    reference's LDMPipelineRange.__call__ (ldm/pipelines.py:282-383) driving the reference-composed Model and the sgm Decoder."""
    from rangeldm_amd.synth import normal
    g = golden("traj")
    cfg = UNetConfig(**SGM_SINUSOID)
    unet = o_unet.OracleUNet(cfg, ref_unet_sd(cfg, "ref/full."))
    vcfg = VAEConfig()
    vae = o_vae.OracleVAE(vcfg, synth_state_dict(vae_param_shapes(vcfg), prefix="vae."))
    x_T = T(normal(51, f"traj/{sched}/x_T", (1, 4, 256, 16)))
    zs = None if sched == "ddim" else [T(normal(52, f"traj/z/{i}", (1, 4, 256, 16))) for i in range(49)] + [None]
    s = o_sched.OracleDDIMScheduler() if sched == "ddim" else o_sched.OracleDDPMScheduler()
    traj = []
    lat = o_pipe.ldm_pipeline(vae, unet, s, x_T, 50, pos_encoding=True, step_noise=zs, trajectory=traj, decode=False)
    ref = T(g[f"traj_{sched}_latent_ref"])
    assert float((lat - ref).norm() / ref.norm()) < 1e-4
    for i in (1, 10, 25, 40, 49):
        r = T(g[f"traj_{sched}_x_step{i}"])
        assert float((traj[i][0] - r).norm() / r.norm()) < 1e-4, i
    img = vae.decode(lat / vcfg.scaling_factor).sample
    r = T(g[f"traj_{sched}_image_ref_f16"]).float()
    assert float((img - r).norm() / r.norm()) < 2e-3


def test_batch16_goldens_pin_the_oracle(golden):
    """tests/golden/b16.npz (oracle/validate_batch16_against_reference.py): the reference-composed full-width Model at the headline
    batch (16 x (5, 256, 16), two timesteps) and 3 DDIM steps of the reference's LDMPipelineRange loop.  Samples never interact,
    so the oracle is re-checked on the first two of the 16 (the CPU suite's time budget); the GPU suite runs all 16."""
    from rangeldm_amd.synth import normal
    g = golden("b16")
    cfg = UNetConfig(**SGM_SINUSOID)
    sd = {k: T(v) for k, v in ref_unet_sd(cfg, "ref/full.").items()}
    x = T(normal(61, "b16/x", (16, 5, 256, 16)))[:2]
    for t in (480, 37):
        ref = T(g[f"b16_eps_t{t}_f16"]).float()[:2]
        eps = o_unet.unet_forward(sd, cfg, x, t)
        assert float((eps - ref).norm() / ref.norm()) < 1e-3              # (the golden is stored as fp16)
    ncfg = UNetConfig(sample_size=(256, 8), **SGM_SINUSOID)          # config 3's per-GPU share: nuScenes latents, 4 images (first 2 here)
    nsd = {k: T(v) for k, v in ref_unet_sd(ncfg, "ref/nusc.").items()}
    ref = T(g["b16_nusc4_eps_t250"])[:2]
    eps = o_unet.unet_forward(nsd, ncfg, T(normal(67, "b16/nusc_x", (4, 5, 256, 8)))[:2], 250)
    assert float((eps - ref).norm() / ref.norm()) < 1e-4
    ref = T(g["b16_nusc32_eps_t610_f16"]).float()[:2]                # ... and of its whole batch of 32 (first 2)
    eps = o_unet.unet_forward(nsd, ncfg, T(normal(68, "b16/nusc32_x", (32, 5, 256, 8)))[:2], 610)
    assert float((eps - ref).norm() / ref.norm()) < 1e-3
    unet = o_unet.OracleUNet(cfg, ref_unet_sd(cfg, "ref/full."))
    x_T = T(normal(62, "b16/x_T", (16, 4, 256, 16)))[:2]
    lat = o_pipe.ldm_pipeline(None, unet, o_sched.OracleDDIMScheduler(), x_T, 3, pos_encoding=True, decode=False)
    ref = T(g["b16_ddim3_latent_f16"]).float()[:2]
    assert float((lat - ref).norm() / ref.norm()) < 1e-3


def test_batch16_headline_golden_pins_the_oracle(golden):
    """tests/golden/b16long.npz: x_0 of the headline workload -- all 50 DDIM steps of the reference's LDMPipelineRange loop at batch 16
    (`python -m oracle.validate_batch16_against_reference --long` checked the oracle on all 16 samples; the CPU suite re-runs the
    first one: samples never interact)."""
    from rangeldm_amd.synth import normal
    g = golden("b16long")
    cfg = UNetConfig(**SGM_SINUSOID)
    unet = o_unet.OracleUNet(cfg, ref_unet_sd(cfg, "ref/full."))
    x_T = T(normal(62, "b16/x_T", (16, 4, 256, 16)))[:1]
    lat = o_pipe.ldm_pipeline(None, unet, o_sched.OracleDDIMScheduler(), x_T, 50, pos_encoding=True, decode=False)
    ref = T(g["b16long_ddim50_latent_f16"]).float()[:1]
    assert float((lat - ref).norm() / ref.norm()) < 1e-3


def test_upscale_full_width_golden_pins_the_oracle(golden):
    """tests/golden/upfull.npz: 10 strided-DDPM steps of the reference's LDMUpscalePipelineRange.__call__ (ldm/pipelines.py:414-519)
    at batch 2 on the full-width 12-channel UNet with SparseRangeImageEncoder2 and the sgm Decoder (BASELINE config 4)."""
    from rangeldm_amd.synth import normal
    g = golden("upfull")
    cfg = UNetConfig(in_channels=12, **SGM_SINUSOID)
    unet = o_unet.OracleUNet(cfg, ref_unet_sd(cfg, "ref/up."))
    vcfg = VAEConfig()
    vae = o_vae.OracleVAE(vcfg, synth_state_dict(vae_param_shapes(vcfg), prefix="vae."))
    cond = o_pipe.sparse_range_image_encoder2(T(normal(63, "upfull/cond", (2, 2, 1024, 16))))
    x_T = T(normal(64, "upfull/x_T", (2, 4, 256, 16)))
    zs = [T(normal(65, f"upfull/z/{i}", (2, 4, 256, 16))) for i in range(9)] + [None]
    lat = o_pipe.ldm_pipeline(vae, unet, o_sched.OracleDDPMScheduler(), x_T, 10, pos_encoding=False, step_noise=zs, cond=cond,
                              decode=False)
    ref = T(g["upfull_latent"])
    assert float((lat - ref).norm() / ref.norm()) < 1e-4
    img = vae.decode(lat / vcfg.scaling_factor).sample
    r = T(g["upfull_image_f16"]).float()
    assert float((img - r).norm() / r.norm()) < 2e-3
