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
