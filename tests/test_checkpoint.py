"""CPU: checkpoint ingestion -- the reference's output_dir layout (ldm/inference.py:46-52,84-127,
ldm/train_unconditional.py:148-177,654-675) and the sgm -> AutoencoderKL conversion (ldm/convert_vae.py:14-189)."""
import json

import numpy as np
import pytest
import torch

from rangeldm_amd import checkpoint as ck
from rangeldm_amd.config import SchedulerConfig, UNetConfig, VAEConfig
from rangeldm_amd.params import unet_param_shapes, vae_param_shapes
from rangeldm_amd.synth import synth_state_dict

SMALL_UNET = UNetConfig(sample_size=(32, 8), block_out_channels=(32, 32, 64, 64))
SMALL_VAE = VAEConfig(ch=32, sample_size=(128, 32))


def test_output_dir_round_trip(tmp_path):
    usd = synth_state_dict(unet_param_shapes(SMALL_UNET), prefix="u.")
    ema = {k: v * 0.5 for k, v in usd.items()}
    vsd = synth_state_dict(vae_param_shapes(SMALL_VAE), prefix="v.")
    ck.save_output_dir(str(tmp_path), SMALL_UNET, usd, SMALL_VAE, vsd, SchedulerConfig(), unet_ema_sd=ema)
    for rel in ("unet/config.json", "unet/diffusion_pytorch_model.safetensors", "vae/config.json",
                "vae/diffusion_pytorch_model.safetensors", "scheduler/scheduler_config.json", "model_index.json",
                "unet_ema/diffusion_pytorch_model.safetensors"):
        assert (tmp_path / rel).exists(), rel
    got = ck.load_output_dir(str(tmp_path))
    assert got["unet_config"] == SMALL_UNET and got["vae_config"] == SMALL_VAE
    assert got["scheduler_config"] == SchedulerConfig()
    assert all(np.array_equal(got["unet"][k], usd[k]) for k in usd) and set(got["unet"]) == set(usd)
    assert all(np.array_equal(got["vae"][k], vsd[k]) for k in vsd)
    assert np.array_equal(ck.load_output_dir(str(tmp_path), ema=True)["unet"]["conv_in.weight"], ema["conv_in.weight"])
    # the files are what `safetensors.torch.load_file` / diffusers expect
    from safetensors.torch import load_file
    t = load_file(str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors"))
    assert t["conv_in.weight"].dtype == torch.float32 and tuple(t["conv_in.weight"].shape) == (32, 5, 3, 3)


def test_reads_a_diffusers_written_config(tmp_path):
    """A config.json with the full set of keys diffusers 0.21-0.26 writes for the RangeLDM UNet."""
    d = {"_class_name": "UNet2DModel", "_diffusers_version": "0.24.0", "act_fn": "silu", "add_attention": True,
         "attention_head_dim": 8, "attn_norm_num_groups": None, "block_out_channels": [128, 128, 256, 256],
         "center_input_sample": False, "class_embed_type": None, "down_block_types":
         ["DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D"], "downsample_padding": 1,
         "downsample_type": "conv", "dropout": 0.0, "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 5,
         "layers_per_block": 2, "mid_block_scale_factor": 1, "norm_eps": 1e-05, "norm_num_groups": 32,
         "num_class_embeds": None, "num_train_timesteps": None, "out_channels": 4, "resnet_time_scale_shift": "default",
         "sample_size": [256, 16], "time_embedding_type": "positional", "up_block_types":
         ["AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D"], "upsample_type": "conv"}
    assert ck.unet_config_from_diffusers(d) == UNetConfig()
    assert ck.unet_config_from_diffusers(json.loads(json.dumps(ck.unet_config_to_diffusers(UNetConfig())))) == UNetConfig()
    for k, v in (("act_fn", "gelu"), ("time_embedding_type", "fourier"), ("resnet_time_scale_shift", "scale_shift"),
                 ("downsample_type", "resnet")):
        with pytest.raises(NotImplementedError, match=k):
            ck.unet_config_from_diffusers({**d, k: v})
    # AutoencoderKL config as ldm/convert_vae.py:123-168 builds it
    v = {"sample_size": [1024, 64], "in_channels": 2, "out_channels": 2, "down_block_types": ["DownEncoderBlock2D"] * 3,
         "up_block_types": ["UpDecoderBlock2D"] * 3, "block_out_channels": [64, 128, 256], "latent_channels": 4,
         "layers_per_block": 2, "scaling_factor": 0.18215}
    assert ck.vae_config_from_diffusers(v) == VAEConfig()
    s = {"_class_name": "DDPMScheduler", "num_train_timesteps": 1000, "beta_start": 0.0001, "beta_end": 0.02,
         "beta_schedule": "linear", "prediction_type": "epsilon", "clip_sample": False, "variance_type": "fixed_small",
         "timestep_spacing": "leading", "steps_offset": 0, "thresholding": False, "trained_betas": None}
    assert ck.scheduler_config_from_diffusers(s) == SchedulerConfig()


def _to_sgm_key(k, levels):
    """Test-side inverse of the key map (diffusers AutoencoderKL -> sgm Encoder/Decoder names)."""
    side, rest = k.split(".", 1)
    p = rest.split(".")
    leaf = {"conv_shortcut": "nin_shortcut", "conv_norm_out": "norm_out"}
    if p[0] in ("conv_in", "conv_out", "conv_norm_out"):
        return f"{side}.{leaf.get(p[0], p[0])}." + ".".join(p[1:])
    if p[0] == "mid_block":
        return f"{side}.mid.block_{int(p[2]) + 1}." + ".".join(leaf.get(x, x) for x in p[3:])
    lvl = int(p[1])
    if p[0] == "up_blocks":
        lvl = levels - 1 - lvl
    top = "down" if p[0] == "down_blocks" else "up"
    if p[2] == "resnets":
        return f"{side}.{top}.{lvl}.block.{p[3]}." + ".".join(leaf.get(x, x) for x in p[4:])
    return f"{side}.{top}.{lvl}.{'downsample' if p[2] == 'downsamplers' else 'upsample'}." + ".".join(p[4:])


def test_sgm_checkpoint_conversion(tmp_path):
    vsd = synth_state_dict(vae_param_shapes(SMALL_VAE), prefix="v.")
    sgm = {"first_stage_model." + _to_sgm_key(k, 3): torch.from_numpy(v) for k, v in vsd.items()}
    sgm["first_stage_model.loss.discriminator.main.0.weight"] = torch.zeros(4)       # dropped (convert_vae.py:17-21)
    sgm["model.diffusion_model.junk"] = torch.zeros(1)                               # not a first-stage key
    assert len(sgm) == len(vsd) + 2
    path = tmp_path / "last.ckpt"
    torch.save({"state_dict": sgm, "epoch": 3}, path)
    y = {"model": {"params": {"encoder_config": {"params": {"attn_type": "none", "double_z": True, "z_channels": 4,
                                                              "in_channels": 2, "out_ch": 2, "ch": 32,
                                                              "ch_mult": [1, 2, 4], "num_res_blocks": 2}}}}}
    import yaml
    (tmp_path / "vae.yaml").write_text(yaml.safe_dump(y))
    cfg, out = ck.load_sgm_vae_checkpoint(str(path), str(tmp_path / "vae.yaml"), image_size=(128, 32))
    assert cfg == SMALL_VAE
    assert set(out) == set(vsd) and all(np.array_equal(np.asarray(out[k]), vsd[k]) for k in vsd)
    bad = dict(sgm)
    bad.pop("first_stage_model.encoder.conv_in.bias")
    torch.save({"state_dict": bad}, path)
    with pytest.raises(RuntimeError, match="missing"):
        ck.load_sgm_vae_checkpoint(str(path), str(tmp_path / "vae.yaml"), image_size=(128, 32))


def test_checkpoint_errors(tmp_path):
    usd = synth_state_dict(unet_param_shapes(SMALL_UNET), prefix="u.")
    ck.save_model_dir(str(tmp_path / "unet"), ck.unet_config_to_diffusers(UNetConfig()), usd)     # config != weights
    with pytest.raises(RuntimeError, match="shape-mismatch"):
        ck.load_unet_dir(str(tmp_path / "unet"))
    with pytest.raises(FileNotFoundError):
        ck.load_output_dir(str(tmp_path / "nope"))
    # a VAE trained WITH mid-block attention must not load as an attention-free one (ldm/inference.py:94-95 swaps attention
    # for identity only when the checkpoint has no attention weights)
    vsd = synth_state_dict(vae_param_shapes(SMALL_VAE), prefix="v.")
    vsd["encoder.mid_block.attentions.0.to_q.weight"] = np.zeros((128, 128), np.float32)
    ck.save_model_dir(str(tmp_path / "vae"), ck.vae_config_to_diffusers(SMALL_VAE), vsd)
    with pytest.raises(NotImplementedError, match="attention"):
        ck.load_vae_dir(str(tmp_path / "vae"))
    # flip_sin_to_cos / freq_shift are UNet2DModel kwargs that change the arithmetic: they round-trip through config.json
    cfg = UNetConfig(sample_size=(32, 8), block_out_channels=(32, 32, 64, 64), flip_sin_to_cos=False, freq_shift=1)
    assert ck.unet_config_from_diffusers(ck.unet_config_to_diffusers(cfg)) == cfg


@pytest.mark.gpu
def test_from_pretrained_round_trip_on_device(tmp_path):
    """save_pretrained -> from_pretrained through the HIP models: same weights, same forward, and the sampling driver
    runs from the directory (`--weights`, ldm/inference.py:46-52)."""
    from rangeldm_amd.pipelines import LDMPipelineRange
    from rangeldm_amd.schedulers import DDPMSchedulerHIP
    from rangeldm_amd.unet import UNet2DModelHIP
    from rangeldm_amd.vae import AutoencoderKLHIP
    ucfg = UNetConfig(sample_size=(256, 16), block_out_channels=(32, 32, 64, 64))
    unet = UNet2DModelHIP(ucfg)
    unet.load_state_dict(synth_state_dict(unet_param_shapes(ucfg), prefix="u."))
    vae = AutoencoderKLHIP(VAEConfig())
    vae.load_state_dict(synth_state_dict(vae_param_shapes(VAEConfig()), prefix="vae."))
    pipe = LDMPipelineRange(vae=vae, unet=unet, scheduler=DDPMSchedulerHIP(), pos_encoding=True)
    pipe.save_pretrained(str(tmp_path))
    u2 = UNet2DModelHIP.from_pretrained(str(tmp_path), subfolder="unet")
    v2 = AutoencoderKLHIP.from_pretrained(str(tmp_path), subfolder="vae")
    s2 = DDPMSchedulerHIP.from_pretrained(str(tmp_path), subfolder="scheduler")
    assert UNet2DModelHIP.load_config(str(tmp_path / "unet" / "config.json")) == ucfg
    assert s2.config.num_train_timesteps == 1000
    x = torch.randn(2, 5, 256, 16, generator=torch.Generator().manual_seed(1)).cuda()
    assert torch.equal(u2(x, 500).sample, unet(x, 500).sample)
    z = torch.randn(1, 4, 256, 16, generator=torch.Generator().manual_seed(2)).cuda()
    assert torch.equal(v2.decode(z).sample, vae.decode(z).sample)
    from rangeldm_amd import inference
    out = tmp_path / "generated"
    inference.main(["--cfg", "RangeLDM", "--weights", str(tmp_path), "--samples", "2", "--batch_size", "2",
                    "--out", str(out)])
    assert sorted(p.name for p in out.iterdir()) == ["0.bin", "0.png", "0_range.png", "1.bin", "1.png", "1_range.png"]
