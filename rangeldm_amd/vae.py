"""AutoencoderKLHIP -- mirror of the diffusers `AutoencoderKL` surface the reference touches:
`vae.decode(z).sample`, `vae.encode(x).latent_dist.sample()`, `vae.config.scaling_factor`.
Accepts diffusers-layout keys (ldm/inference.py:97) or sgm `AutoencodingEngine` keys (mapped as ldm/convert_vae.py does)."""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from .config import VAEConfig
from .params import vae_param_shapes, sgm_to_diffusers_vae_key
from .unet import _as_host_f32


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class DiagonalGaussianDistributionHIP:
    """vae/sgm/modules/distributions/distributions.py:24-41 on device moments."""

    def __init__(self, moments):
        self.parameters = moments
        self.mean, lv = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(lv, -30.0, 20.0)

    def sample(self, generator=None, noise=None, scale=1.0):
        m = self.parameters
        B, z2, w, h = m.shape
        if noise is None:
            # reference draws on the CPU then moves (`torch.randn(shape).to(device)`, distributions.py:38-41)
            noise = torch.randn((B, z2 // 2, w, h), generator=generator)
        noise = noise.to(device=m.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(noise)
        _lib.check(_lib.lib().rldm_diag_gaussian_sample(C.c_void_p(m.data_ptr()), C.c_void_p(noise.data_ptr()),
                                                        float(scale), B, z2 // 2, w * h, C.c_void_p(out.data_ptr()),
                                                        _lib.stream_ptr(m.device)), "rldm_diag_gaussian_sample")
        return out

    def mode(self):
        return self.mean


class EncoderOutput:
    def __init__(self, dist):
        self.latent_dist = dist


class AutoencoderKLHIP:
    def __init__(self, config=None, device="cuda", **kwargs):
        if config is None:
            config = VAEConfig(**kwargs)
        elif isinstance(config, dict):
            config = VAEConfig(**config)
        self._cfg = config
        self.config = SimpleNamespace(**config.to_dict(), latent_channels=config.z_channels)
        self.device = torch.device(device)
        self.dtype = torch.float32
        _lib.require_gpu()
        c = _lib.VAEConfigC()
        c.in_channels, c.out_channels, c.ch = config.in_channels, config.out_channels, config.ch
        c.num_levels = len(config.ch_mult)
        for i, v in enumerate(config.ch_mult):
            c.ch_mult[i] = v
        c.num_res_blocks, c.z_channels = config.num_res_blocks, config.z_channels
        c.double_z = 1 if config.double_z else 0
        c.norm_num_groups, c.norm_eps, c.scaling_factor = config.norm_num_groups, config.norm_eps, config.scaling_factor
        self._h = C.c_void_p()
        _lib.check(_lib.lib().rldm_vae_create(C.byref(c), C.byref(self._h)), "rldm_vae_create")
        self._shapes = vae_param_shapes(config)
        self._state = {}
        self._finalized = False

    @classmethod
    def from_config(cls, config, **kw):
        return cls(config, **kw)

    @classmethod
    def load_config(cls, path, subfolder=None):
        """`AutoencoderKL.load_config(args.vae_config)` (ldm/inference.py:86)."""
        import json
        import os
        from .checkpoint import CONFIG_NAME, vae_config_from_diffusers
        if subfolder:
            path = os.path.join(path, subfolder)
        if os.path.isdir(path):
            path = os.path.join(path, CONFIG_NAME)
        with open(path) as f:
            return vae_config_from_diffusers(json.load(f))

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        import os
        from .checkpoint import load_vae_dir
        cfg, sd = load_vae_dir(os.path.join(path, subfolder) if subfolder else path)
        m = cls(cfg, **kw)
        m.load_state_dict(sd)
        return m

    @classmethod
    def from_sgm_checkpoint(cls, ckpt_path, yaml_path=None, image_size=None, **kw):
        """ldm/convert_vae.py:149-189: an sgm AutoencodingEngine `.ckpt` (+ its yaml) straight into the HIP VAE."""
        from .checkpoint import load_sgm_vae_checkpoint
        cfg, sd = load_sgm_vae_checkpoint(ckpt_path, yaml_path, image_size)
        m = cls(cfg, **kw)
        m.load_state_dict(sd)
        return m

    def save_pretrained(self, path):
        from .checkpoint import save_model_dir, vae_config_to_diffusers
        save_model_dir(path, vae_config_to_diffusers(self._cfg), self._state)

    def load_state_dict(self, state_dict, strict=True):
        sd = {}
        for k, v in state_dict.items():
            if k in self._shapes:
                sd[k] = v
            else:
                dk = sgm_to_diffusers_vae_key(k, len(self._cfg.ch_mult))
                if dk is not None and dk in self._shapes:
                    sd[dk] = v
                elif ".attentions." in k or ".attn_" in k or ".attn." in k:
                    # ldm/inference.py:94-95 swaps attention for identity only when the checkpoint has NO attention
                    # weights; one that has them was trained with mid-block attention and would decode wrongly here
                    raise NotImplementedError(f"VAE checkpoint holds attention weights ({k}): mid-block attention is not "
                                              "implemented (the reference's range-image VAEs use attn_type none)")
                elif strict and (k.startswith("encoder.") or k.startswith("decoder.")):
                    raise RuntimeError(f"unexpected VAE key {k}")
        missing = [k for k in self._shapes if k not in sd]
        if strict and missing:
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]}")
        L = _lib.lib()
        for k, shape in self._shapes.items():
            if k not in sd:
                continue
            a = _as_host_f32(sd[k]).reshape(shape)
            self._state[k] = a
            _lib.check(L.rldm_vae_set_param(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), a.size), f"set_param {k}")
        _lib.check(L.rldm_vae_finalize(self._h), "rldm_vae_finalize")
        self._finalized = True
        return SimpleNamespace(missing_keys=missing, unexpected_keys=[])

    def state_dict(self):
        return {k: torch.from_numpy(v.copy()) for k, v in self._state.items()}

    def to(self, *a, **k):
        return self

    def decode(self, z, return_dict=True):
        if not self._finalized:
            raise RuntimeError("AutoencoderKLHIP: load_state_dict has not been called")
        z = z.to(device=self.device, dtype=torch.float32).contiguous()
        B, zc, w, h = z.shape
        if zc != self._cfg.z_channels:
            raise ValueError(f"latent has {zc} channels, VAE expects {self._cfg.z_channels}")
        f = self._cfg.downscale
        img = torch.empty((B, self._cfg.out_channels, w * f, h * f), device=self.device, dtype=torch.float32)
        _lib.check(_lib.lib().rldm_vae_decode(self._h, C.c_void_p(z.data_ptr()), B, w, h, C.c_void_p(img.data_ptr()),
                                              _lib.stream_ptr(self.device)), "rldm_vae_decode")
        return DecoderOutput(img) if return_dict else (img,)

    def encode(self, x, return_dict=True):
        if not self._finalized:
            raise RuntimeError("AutoencoderKLHIP: load_state_dict has not been called")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B, c, w, h = x.shape
        f = self._cfg.downscale
        zc = self._cfg.z_channels * (2 if self._cfg.double_z else 1)
        mom = torch.empty((B, zc, w // f, h // f), device=self.device, dtype=torch.float32)
        _lib.check(_lib.lib().rldm_vae_encode(self._h, C.c_void_p(x.data_ptr()), B, w, h, C.c_void_p(mom.data_ptr()),
                                              _lib.stream_ptr(self.device)), "rldm_vae_encode")
        return EncoderOutput(DiagonalGaussianDistributionHIP(mom))

    def decode_flops(self, batch, latent_w, latent_h):
        return float(_lib.lib().rldm_vae_decode_flops(self._h, batch, latent_w, latent_h))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().rldm_vae_destroy(self._h)
                self._h = None
        except Exception:
            pass
