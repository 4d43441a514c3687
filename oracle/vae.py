"""Oracle VAE (fp32 torch).  Arithmetic follows sgm Encoder.forward / Decoder.forward
(vae/sgm/modules/diffusionmodules/model.py:852-896, 1024-1057) with circular=True, attn_type none, act silu
(vae/configs/kitti360.yaml:30-62); parameter names are the diffusers AutoencoderKL ones the reference runs after
ldm/convert_vae.py (key map: rangeldm_amd.params.sgm_to_diffusers_vae_key)."""
import torch
from rangeldm_amd.config import VAEConfig
from . import ops
from .unet import resnet_block, _t


@torch.no_grad()
def vae_encode(sd, cfg: VAEConfig, x):
    """(B,2,W,H) -> moments (B, 2*z, W/4, H/4)  [mean | logvar]."""
    G, eps, L = cfg.norm_num_groups, cfg.norm_eps, len(cfg.ch_mult)
    h = ops.circ_conv2d(x, _t(sd, "encoder.conv_in.weight"), _t(sd, "encoder.conv_in.bias"))
    for i in range(L):
        for j in range(cfg.num_res_blocks):
            h = resnet_block(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, None, G, eps)
        if i != L - 1:
            p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            h = ops.downsample_vae(h, _t(sd, p + ".weight"), _t(sd, p + ".bias"))
    h = resnet_block(sd, "encoder.mid_block.resnets.0", h, None, G, eps)
    h = resnet_block(sd, "encoder.mid_block.resnets.1", h, None, G, eps)
    h = ops.group_norm_silu(h, _t(sd, "encoder.conv_norm_out.weight"), _t(sd, "encoder.conv_norm_out.bias"), G, eps)
    return ops.circ_conv2d(h, _t(sd, "encoder.conv_out.weight"), _t(sd, "encoder.conv_out.bias"))


@torch.no_grad()
def vae_decode(sd, cfg: VAEConfig, z):
    """(B, z, W/4, H/4) -> (B, 2, W, H)."""
    G, eps, L = cfg.norm_num_groups, cfg.norm_eps, len(cfg.ch_mult)
    h = ops.circ_conv2d(z, _t(sd, "decoder.conv_in.weight"), _t(sd, "decoder.conv_in.bias"))
    h = resnet_block(sd, "decoder.mid_block.resnets.0", h, None, G, eps)
    h = resnet_block(sd, "decoder.mid_block.resnets.1", h, None, G, eps)
    for i in range(L):
        for j in range(cfg.num_res_blocks + 1):
            h = resnet_block(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, None, G, eps)
        if i != L - 1:
            p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            h = ops.upsample_conv(h, _t(sd, p + ".weight"), _t(sd, p + ".bias"))
    h = ops.group_norm_silu(h, _t(sd, "decoder.conv_norm_out.weight"), _t(sd, "decoder.conv_norm_out.bias"), G, eps)
    return ops.circ_conv2d(h, _t(sd, "decoder.conv_out.weight"), _t(sd, "decoder.conv_out.bias"))


class DiagonalGaussian:
    """vae/sgm/modules/distributions/distributions.py:24-41."""

    def __init__(self, moments):
        self.mean, logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, noise=None, generator=None):
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class OracleVAE:
    """Duck-types vae.decode(z).sample / vae.encode(x).latent_dist.sample() / vae.config.scaling_factor."""

    class _Dec:
        def __init__(self, s):
            self.sample = s

    class _Enc:
        def __init__(self, d):
            self.latent_dist = d

    def __init__(self, cfg: VAEConfig, state_dict):
        self.config = cfg
        self.sd = {k: (v if torch.is_tensor(v) else torch.from_numpy(v)).float() for k, v in state_dict.items()}

    def decode(self, z):
        return self._Dec(vae_decode(self.sd, self.config, z.float()))

    def encode(self, x):
        return self._Enc(DiagonalGaussian(vae_encode(self.sd, self.config, x.float())))
