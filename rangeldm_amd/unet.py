"""UNet2DModelHIP -- host-side mirror of the diffusers `UNet2DModel` surface the reference touches
(SURVEY.md 8b): `unet(sample, timestep).sample`, `.config.{sample_size,in_channels,out_channels}`, `.dtype`,
`.device`, `load_state_dict`, `state_dict`.  All arithmetic runs in librangeldm_hip (no torch ops on the path)."""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from .config import UNetConfig
from .params import unet_param_shapes


class UNet2DOutput:
    """diffusers.models.unet_2d.UNet2DOutput stand-in: `.sample`."""

    def __init__(self, sample):
        self.sample = sample


def _as_host_f32(v):
    if torch.is_tensor(v):
        v = v.detach().to("cpu", torch.float32).numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


class UNet2DModelHIP:
    def __init__(self, config=None, device="cuda", **kwargs):
        """UNet2DModelHIP(**model_config) as ldm/train_unconditional.py:237-242 calls UNet2DModel(**model_config)."""
        if config is None:
            config = UNetConfig(**kwargs)
        elif isinstance(config, dict):
            config = UNetConfig(**config)
        self._cfg = config
        self.config = SimpleNamespace(**config.to_dict())
        self.device = torch.device(device)
        self.dtype = torch.float32            # boundary dtype (ldm/pipelines.py:224 draws noise in unet.dtype)
        _lib.require_gpu()
        L = _lib.lib()
        c = _lib.UNetConfigC()
        c.sample_w, c.sample_h = config.sample_size
        c.in_channels, c.out_channels = config.in_channels, config.out_channels
        c.layers_per_block = config.layers_per_block
        c.num_levels = len(config.block_out_channels)
        for i, v in enumerate(config.block_out_channels):
            c.block_out_channels[i] = v
            c.down_attn[i] = 1 if config.down_block_types[i] == "AttnDownBlock2D" else 0
            c.up_attn[i] = 1 if config.up_block_types[i] == "AttnUpBlock2D" else 0
        for t in config.down_block_types:
            if t not in ("DownBlock2D", "AttnDownBlock2D"):
                raise NotImplementedError(f"down block type {t}")
        for t in config.up_block_types:
            if t not in ("UpBlock2D", "AttnUpBlock2D"):
                raise NotImplementedError(f"up block type {t}")
        c.attention_head_dim = config.attention_head_dim
        c.norm_num_groups = config.norm_num_groups
        c.norm_eps = config.norm_eps
        c.mid_attention = 1 if config.add_attention else 0
        c.flip_sin_to_cos, c.freq_shift = int(bool(config.flip_sin_to_cos)), int(config.freq_shift)
        self._h = C.c_void_p()
        _lib.check(L.rldm_unet_create(C.byref(c), C.byref(self._h)), "rldm_unet_create")
        self._shapes = unet_param_shapes(config)
        self._state = {}
        self._finalized = False

    @classmethod
    def from_config(cls, config, **kw):
        return cls(config, **kw)

    @classmethod
    def load_config(cls, path, subfolder=None):
        """`UNet2DModel.load_config(args.unet_config)` (ldm/inference.py:84): a config.json path or its directory."""
        import json
        import os
        from .checkpoint import CONFIG_NAME, unet_config_from_diffusers
        if subfolder:
            path = os.path.join(path, subfolder)
        if os.path.isdir(path):
            path = os.path.join(path, CONFIG_NAME)
        with open(path) as f:
            return unet_config_from_diffusers(json.load(f))

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        """`UNet2DModel.from_pretrained(input_dir, subfolder="unet")` (ldm/train_unconditional.py:169)."""
        import os
        from .checkpoint import load_unet_dir
        cfg, sd = load_unet_dir(os.path.join(path, subfolder) if subfolder else path)
        m = cls(cfg, **kw)
        m.load_state_dict(sd)
        return m

    def save_pretrained(self, path):
        """`model.save_pretrained(os.path.join(output_dir, "unet"))` (ldm/train_unconditional.py:152)."""
        from .checkpoint import save_model_dir, unet_config_to_diffusers
        save_model_dir(path, unet_config_to_diffusers(self._cfg), self._state)

    # -- weights ------------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        missing = [k for k in self._shapes if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        L = _lib.lib()
        for k, shape in self._shapes.items():
            if k not in state_dict:
                continue
            a = _as_host_f32(state_dict[k])
            if tuple(a.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(a.shape)} vs model {tuple(shape)}")
            self._state[k] = a
            _lib.check(L.rldm_unet_set_param(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), a.size), f"set_param {k}")
        _lib.check(L.rldm_unet_finalize(self._h), "rldm_unet_finalize")
        self._finalized = True
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def state_dict(self):
        return {k: torch.from_numpy(v.copy()) for k, v in self._state.items()}

    def parameters(self):
        return iter(self.state_dict().values())

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    # -- forward ------------------------------------------------------------------------------------------------
    def __call__(self, sample, timestep, return_dict=True):
        """sample (B, in_channels, W, H) on the GPU; timestep int / 0-d / 1-d (len B) tensor."""
        if not self._finalized:
            raise RuntimeError("UNet2DModelHIP: load_state_dict has not been called")
        x = sample.to(device=self.device, dtype=torch.float32).contiguous()
        B = x.shape[0]
        cfg = self._cfg
        if tuple(x.shape[1:]) != (cfg.in_channels, *cfg.sample_size):
            raise ValueError(f"sample shape {tuple(x.shape)} != (B, {cfg.in_channels}, {cfg.sample_size})")
        if torch.is_tensor(timestep):
            t = timestep.detach().to("cpu", torch.int64).reshape(-1).numpy()
        else:
            t = np.asarray([timestep], dtype=np.int64)
        t = np.ascontiguousarray(t, dtype=np.int64)
        if t.size not in (1, B):
            raise ValueError("timestep must be a scalar or have one entry per sample")
        out = torch.empty((B, cfg.out_channels, *cfg.sample_size), device=self.device, dtype=torch.float32)
        _lib.check(_lib.lib().rldm_unet_forward(self._h, C.c_void_p(x.data_ptr()), t.ctypes.data_as(C.c_void_p),
                                                int(t.size), B, C.c_void_p(out.data_ptr()),
                                                _lib.stream_ptr(self.device)), "rldm_unet_forward")
        return UNet2DOutput(out) if return_dict else (out,)

    def flops(self, batch):
        return float(_lib.lib().rldm_unet_flops(self._h, batch))

    def num_launches(self, batch):
        return int(_lib.lib().rldm_unet_num_launches(self._h, batch))

    def trunk_status(self, batch):
        """Self-check of the persistent trunk launches of the batch's plan (0: fine or none); synchronises the device."""
        return int(_lib.lib().rldm_unet_trunk_status(self._h, batch))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().rldm_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass
