"""Checkpoint ingestion: the on-disk layouts the reference reads and writes.

  <output_dir>/unet/config.json + diffusion_pytorch_model.safetensors        ldm/inference.py:46-47,84-85,120
  <output_dir>/vae/config.json  + diffusion_pytorch_model.safetensors        ldm/inference.py:50-51,86-97
  <output_dir>/scheduler/scheduler_config.json                               ldm/inference.py:48,126-127
  <output_dir>/unet_ema/...  (EMA copy written by the training hook)         ldm/train_unconditional.py:148-153
  sgm AutoencodingEngine `.ckpt` (torch pickle, {"state_dict": ...}) + its yaml -> AutoencoderKL keys / config
                                                                             ldm/convert_vae.py:14-189

Host-side only: json / safetensors / key renaming.  The tensors end up in librangeldm_hip through
`UNet2DModelHIP.load_state_dict` / `AutoencoderKLHIP.load_state_dict`; nothing here touches the GPU.
"""
import json
import os

from .config import SchedulerConfig, UNetConfig, VAEConfig
from .params import sgm_to_diffusers_vae_key, unet_param_shapes, vae_param_shapes

WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
CONFIG_NAME = "config.json"
SCHEDULER_CONFIG_NAME = "scheduler_config.json"

# diffusers UNet2DModel defaults the kernels implement; any other value in a config.json is refused loudly
_UNET_FIXED = {
    "act_fn": "silu", "time_embedding_type": "positional",
    "resnet_time_scale_shift": "default", "downsample_type": "conv", "upsample_type": "conv", "dropout": 0.0,
    "mid_block_scale_factor": 1, "downsample_padding": 1, "center_input_sample": False, "class_embed_type": None,
    "num_class_embeds": None, "attn_norm_num_groups": None,
}


def _read_json(path):
    with open(path) as f:
        return json.load(f)


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else tuple(int(x) for x in v)


# ---- config.json <-> dataclasses ---------------------------------------------------------------------------------
def unet_config_from_diffusers(d):
    """UNet2DModel config.json (written by `save_pretrained`, ldm/train_unconditional.py:152) -> UNetConfig."""
    for k, want in _UNET_FIXED.items():
        if k in d and d[k] != want:
            raise NotImplementedError(f"UNet2DModel config {k}={d[k]!r}: only {want!r} is implemented")
    if d.get("_class_name", "UNet2DModel") != "UNet2DModel":
        raise ValueError(f"not a UNet2DModel config: _class_name={d.get('_class_name')!r}")
    kw = dict(sample_size=_pair(d["sample_size"]), in_channels=d["in_channels"], out_channels=d["out_channels"],
              layers_per_block=d.get("layers_per_block", 2), block_out_channels=tuple(d["block_out_channels"]),
              down_block_types=tuple(d["down_block_types"]), up_block_types=tuple(d["up_block_types"]),
              attention_head_dim=d.get("attention_head_dim", 8) or 8, norm_num_groups=d.get("norm_num_groups", 32),
              norm_eps=d.get("norm_eps", 1e-5), add_attention=bool(d.get("add_attention", True)),
              flip_sin_to_cos=bool(d.get("flip_sin_to_cos", True)), freq_shift=int(d.get("freq_shift", 0)))
    return UNetConfig(**kw)


def unet_config_to_diffusers(cfg):
    d = {"_class_name": "UNet2DModel", "_diffusers_version": "0.21.0", "sample_size": list(cfg.sample_size),
         "in_channels": cfg.in_channels, "out_channels": cfg.out_channels, "layers_per_block": cfg.layers_per_block,
         "block_out_channels": list(cfg.block_out_channels), "down_block_types": list(cfg.down_block_types),
         "up_block_types": list(cfg.up_block_types), "attention_head_dim": cfg.attention_head_dim,
         "norm_num_groups": cfg.norm_num_groups, "norm_eps": cfg.norm_eps,
         "flip_sin_to_cos": bool(cfg.flip_sin_to_cos), "freq_shift": int(cfg.freq_shift)}
    d.update(_UNET_FIXED)
    d["add_attention"] = bool(cfg.add_attention)
    return d


def vae_config_from_diffusers(d):
    """AutoencoderKL config.json as ldm/convert_vae.py:123-168 builds it -> VAEConfig (sgm Encoder/Decoder kwargs)."""
    boc = list(d["block_out_channels"])
    ch = boc[0]
    if any(c % ch for c in boc):
        raise ValueError(f"block_out_channels {boc} are not multiples of the base width {ch}")
    for k in ("down_block_types", "up_block_types"):
        want = "DownEncoderBlock2D" if k.startswith("down") else "UpDecoderBlock2D"
        if any(t != want for t in d.get(k, [want])):
            raise NotImplementedError(f"AutoencoderKL {k}={d[k]!r}: only {want} is implemented")
    if d.get("act_fn", "silu") != "silu":
        raise NotImplementedError(f"AutoencoderKL act_fn={d['act_fn']!r}")
    kw = dict(in_channels=d.get("in_channels", 2), out_channels=d.get("out_channels", 2), ch=ch,
              ch_mult=tuple(c // ch for c in boc), num_res_blocks=d.get("layers_per_block", 2),
              z_channels=d.get("latent_channels", 4), norm_num_groups=d.get("norm_num_groups", 32),
              scaling_factor=d.get("scaling_factor", 0.18215))
    if "sample_size" in d and d["sample_size"] is not None:
        kw["sample_size"] = _pair(d["sample_size"])
    return VAEConfig(**kw)


def vae_config_to_diffusers(cfg):
    n = len(cfg.ch_mult)
    return {"_class_name": "AutoencoderKL", "_diffusers_version": "0.21.0", "sample_size": list(cfg.sample_size),
            "in_channels": cfg.in_channels, "out_channels": cfg.out_channels,
            "down_block_types": ["DownEncoderBlock2D"] * n, "up_block_types": ["UpDecoderBlock2D"] * n,
            "block_out_channels": [cfg.ch * m for m in cfg.ch_mult], "latent_channels": cfg.z_channels,
            "layers_per_block": cfg.num_res_blocks, "act_fn": "silu", "norm_num_groups": cfg.norm_num_groups,
            "scaling_factor": cfg.scaling_factor}


def vae_config_from_sgm_yaml(y, image_size=None):
    """The sgm training yaml (vae/configs/kitti360.yaml:30-62) -> VAEConfig, as create_vae_diffusers_config reads it:
    `model.params.ddconfig` or `model.params.encoder_config.params` (ldm/convert_vae.py:127-131)."""
    p = y["model"]["params"]
    dd = p["ddconfig"] if "ddconfig" in p else p["encoder_config"]["params"]
    if dd.get("attn_type", "none") != "none":
        raise NotImplementedError("VAE attention (attn_type != none): the reference's range-image VAEs are attention-free")
    kw = dict(in_channels=dd["in_channels"], out_channels=dd["out_ch"], ch=dd["ch"], ch_mult=tuple(dd["ch_mult"]),
              num_res_blocks=dd["num_res_blocks"], z_channels=dd["z_channels"], double_z=bool(dd.get("double_z", True)),
              scaling_factor=float(p.get("scale_factor", 0.18215)))
    if image_size is not None:
        kw["sample_size"] = _pair(image_size)
    return VAEConfig(**kw)


def scheduler_config_from_diffusers(d):
    return SchedulerConfig(**{k: v for k, v in d.items() if k in SchedulerConfig.__dataclass_fields__})


def scheduler_config_to_diffusers(cfg, class_name="DDPMScheduler"):
    d = {"_class_name": class_name, "_diffusers_version": "0.21.0"}
    d.update(cfg.to_dict())
    return d


# ---- state dicts -------------------------------------------------------------------------------------------------
def _load_safetensors(path):
    """fp32 numpy state dict (released checkpoints may be stored in fp16 / bf16: read through torch, widen)."""
    from safetensors.torch import load_file
    return {k: v.float().numpy() for k, v in load_file(path).items()}


def _save_safetensors(sd, path):
    import numpy as np
    from safetensors.numpy import save_file
    out = {}
    for k, v in sd.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        out[k] = np.ascontiguousarray(v, dtype=np.float32)
    save_file(out, path)


def check_state_dict(sd, shapes, what):
    missing = [k for k in shapes if k not in sd]
    unexpected = [k for k in sd if k not in shapes]
    bad = [k for k in shapes if k in sd and tuple(sd[k].shape) != tuple(shapes[k])]
    if missing or unexpected or bad:
        raise RuntimeError(f"{what}: missing {missing[:4]} unexpected {unexpected[:4]} shape-mismatch {bad[:4]} "
                           f"({len(missing)}/{len(unexpected)}/{len(bad)})")


def convert_sgm_vae_state_dict(checkpoint, num_levels=3):
    """ldm/convert_vae.py:14-121: strip `first_stage_model.`, rename encoder / decoder keys, drop loss / discriminator
    keys; quant convs are kept only if present (the range-image VAEs have none, :167-168)."""
    keys = list(checkpoint.keys())
    prefix = "first_stage_model." if any(k.startswith("first_stage_model.") for k in keys) else ""
    out = {}
    for k in keys:
        if not k.startswith(prefix):
            continue
        kk = k[len(prefix):]
        if kk.startswith("quant_conv.") or kk.startswith("post_quant_conv."):
            out[kk] = checkpoint[k]
            continue
        dk = sgm_to_diffusers_vae_key(kk, num_levels)
        if dk is not None:
            out[dk] = checkpoint[k]
    return out


def load_sgm_vae_checkpoint(ckpt_path, yaml_path=None, image_size=None):
    """sgm `.ckpt` (+ yaml) -> (VAEConfig, diffusers-keyed state dict): what ldm/convert_vae.py:149-189 produces."""
    import torch
    blob = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    sd = blob["state_dict"] if "state_dict" in blob else blob
    cfg = VAEConfig() if image_size is None else VAEConfig(sample_size=_pair(image_size))
    if yaml_path is not None:
        import yaml
        with open(yaml_path) as f:
            cfg = vae_config_from_sgm_yaml(yaml.safe_load(f), image_size)
    out = convert_sgm_vae_state_dict(sd, len(cfg.ch_mult))
    if any(k.startswith("quant_conv.") for k in out):
        raise NotImplementedError("VAE with quant_conv / post_quant_conv (the reference replaces them by Identity)")
    check_state_dict(out, vae_param_shapes(cfg), f"sgm checkpoint {ckpt_path}")
    return cfg, out


# ---- the reference's output_dir ----------------------------------------------------------------------------------
def load_unet_dir(path):
    """`UNet2DModel.from_pretrained(dir, subfolder="unet")` counterpart: (UNetConfig, numpy state dict)."""
    cfg = unet_config_from_diffusers(_read_json(os.path.join(path, CONFIG_NAME)))
    sd = _load_safetensors(os.path.join(path, WEIGHTS_NAME))
    check_state_dict(sd, unet_param_shapes(cfg), f"UNet checkpoint {path}")
    return cfg, sd


def load_vae_dir(path):
    cfg = vae_config_from_diffusers(_read_json(os.path.join(path, CONFIG_NAME)))
    sd = _load_safetensors(os.path.join(path, WEIGHTS_NAME))
    if "quant_conv.weight" in sd:                     # ldm/inference.py:89-91 keeps them only when present
        raise NotImplementedError("VAE checkpoint with quant_conv / post_quant_conv")
    attn = [k for k in sd if ".attentions." in k]
    if attn:
        # ldm/inference.py:94-95 replaces attention by identity only when 'encoder.mid_block.attentions.0.to_q.weight' is
        # ABSENT; a checkpoint that carries attention weights was trained with it and must not be run attention-free
        raise NotImplementedError(f"VAE checkpoint {path} holds mid-block attention weights ({attn[0]}, ...): not implemented")
    check_state_dict(sd, vae_param_shapes(cfg), f"VAE checkpoint {path}")
    return cfg, sd


def load_scheduler_dir(path):
    return scheduler_config_from_diffusers(_read_json(os.path.join(path, SCHEDULER_CONFIG_NAME)))


def load_output_dir(output_dir, with_vae=True, ema=False):
    """Everything ldm/inference.py:46-52 derives from `output_dir`.  ema=True reads `unet_ema/` (the EMA weights the
    training hook stores next to `unet/`)."""
    out = {}
    out["unet_config"], out["unet"] = load_unet_dir(os.path.join(output_dir, "unet_ema" if ema else "unet"))
    sched = os.path.join(output_dir, "scheduler")
    out["scheduler_config"] = load_scheduler_dir(sched) if os.path.isdir(sched) else SchedulerConfig()
    if with_vae:
        out["vae_config"], out["vae"] = load_vae_dir(os.path.join(output_dir, "vae"))
    return out


def save_model_dir(path, config_dict, state_dict):
    """`model.save_pretrained(path)` counterpart: config.json + diffusion_pytorch_model.safetensors."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, CONFIG_NAME), "w") as f:
        json.dump(config_dict, f, indent=2, sort_keys=True)
    _save_safetensors(state_dict, os.path.join(path, WEIGHTS_NAME))


def save_output_dir(output_dir, unet_cfg, unet_sd, vae_cfg=None, vae_sd=None, scheduler_cfg=None, unet_ema_sd=None):
    """`pipeline.save_pretrained(args.output_dir)` (ldm/train_unconditional.py:654-675) + the `unet_ema/` of the hook."""
    save_model_dir(os.path.join(output_dir, "unet"), unet_config_to_diffusers(unet_cfg), unet_sd)
    if unet_ema_sd is not None:
        save_model_dir(os.path.join(output_dir, "unet_ema"), unet_config_to_diffusers(unet_cfg), unet_ema_sd)
    if vae_cfg is not None:
        save_model_dir(os.path.join(output_dir, "vae"), vae_config_to_diffusers(vae_cfg), vae_sd)
    sdir = os.path.join(output_dir, "scheduler")
    os.makedirs(sdir, exist_ok=True)
    with open(os.path.join(sdir, SCHEDULER_CONFIG_NAME), "w") as f:
        json.dump(scheduler_config_to_diffusers(scheduler_cfg or SchedulerConfig()), f, indent=2, sort_keys=True)
    index = {"_class_name": "LDMPipelineRange" if vae_cfg is not None else "DDIMPipelineRange",
             "_diffusers_version": "0.21.0", "unet": ["diffusers", "UNet2DModel"],
             "scheduler": ["diffusers", "DDPMScheduler"]}
    if vae_cfg is not None:
        index["vae"] = ["diffusers", "AutoencoderKL"]
    with open(os.path.join(output_dir, "model_index.json"), "w") as f:
        json.dump(index, f, indent=2, sort_keys=True)
