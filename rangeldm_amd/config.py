"""Configurations of the RangeLDM hot path (UNet2DModel / AutoencoderKL / scheduler).

Mirrors the keyword surface the reference passes to diffusers:
  * UNet2DModel(**model_config)         -- /root/reference ldm/train_unconditional.py:237-242,
                                            ldm/configs/{RangeLDM,RangeDM,nuscenes,upsample}.yaml `model_config`
  * AutoencoderKL via convert_vae        -- ldm/convert_vae.py:123-147, vae/configs/kitti360.yaml:30-62
  * DDPMScheduler(...)                   -- ldm/train_unconditional.py:347-352

Library defaults that the reference relies on (diffusers 0.21-0.26 UNet2DModel.__init__) are spelled out here
because the hot path depends on them: attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5, silu,
positional time embedding with flip_sin_to_cos=True / freq_shift=0, downsample_padding=1.
"""
from dataclasses import dataclass, field, asdict
from typing import Tuple


@dataclass
class UNetConfig:
    sample_size: Tuple[int, int] = (256, 16)          # (W azimuth, H beams) of the tensor the UNet sees
    in_channels: int = 5
    out_channels: int = 4
    layers_per_block: int = 2
    block_out_channels: Tuple[int, ...] = (128, 128, 256, 256)
    down_block_types: Tuple[str, ...] = ("DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D")
    up_block_types: Tuple[str, ...] = ("AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D")
    attention_head_dim: int = 8
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    time_embed_dim_mult: int = 4                       # time_embed_dim = block_out_channels[0] * 4
    add_attention: bool = True                         # mid-block attention
    # diffusers UNet2DModel kwargs of the positional time embedding (Timesteps(block_out_channels[0], flip_sin_to_cos,
    # freq_shift)); (False, 1) is the sinusoid of sgm's get_timestep_embedding (vae/sgm/.../model.py:28-46)
    flip_sin_to_cos: bool = True
    freq_shift: int = 0
    downsample_padding: int = 1                        # the only value the reference's UNets use (ldm/utils.py:92-93)
    # reference surgery (ldm/utils.py:125-203 via `all_circonv`): every conv wraps W and zero-pads H
    all_circonv: bool = True

    def __post_init__(self):
        self.sample_size = tuple(self.sample_size)
        self.block_out_channels = tuple(self.block_out_channels)
        self.down_block_types = tuple(self.down_block_types)
        self.up_block_types = tuple(self.up_block_types)
        assert len(self.block_out_channels) == len(self.down_block_types) == len(self.up_block_types)
        if self.downsample_padding != 1:
            raise NotImplementedError("UNet downsamplers use downsample_padding=1 (circular 3x3 stride 2)")
        if not self.all_circonv:
            raise NotImplementedError("only the all_circonv surgery of the reference configs is supported "
                                      "(sub_circonv, ldm/inference.py:105-118, is out of scope)")

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * self.time_embed_dim_mult

    def to_dict(self):
        return asdict(self)


@dataclass
class VAEConfig:
    """sgm Encoder/Decoder kwargs (vae/configs/kitti360.yaml:30-62) == diffusers AutoencoderKL after convert_vae."""
    in_channels: int = 2
    out_channels: int = 2
    ch: int = 64
    ch_mult: Tuple[int, ...] = (1, 2, 4)
    num_res_blocks: int = 2
    z_channels: int = 4
    double_z: bool = True
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215                    # ldm/convert_vae.py:159-168
    sample_size: Tuple[int, int] = (1024, 64)

    def __post_init__(self):
        self.ch_mult = tuple(self.ch_mult)
        self.sample_size = tuple(self.sample_size)

    @property
    def latent_channels(self):
        return self.z_channels

    @property
    def downscale(self):
        return 2 ** (len(self.ch_mult) - 1)

    def to_dict(self):
        return asdict(self)


@dataclass
class SchedulerConfig:
    """diffusers DDPMScheduler config as built by ldm/train_unconditional.py:347-352."""
    num_train_timesteps: int = 1000
    beta_start: float = 1e-4
    beta_end: float = 0.02
    beta_schedule: str = "linear"
    prediction_type: str = "epsilon"
    clip_sample: bool = False
    variance_type: str = "fixed_small"
    timestep_spacing: str = "leading"
    steps_offset: int = 0
    set_alpha_to_one: bool = True

    def to_dict(self):
        return asdict(self)


_D, _A = "DownBlock2D", "AttnDownBlock2D"
_U, _AU = "UpBlock2D", "AttnUpBlock2D"

PRESETS = {
    # ldm/configs/RangeLDM.yaml:17-24
    "RangeLDM": dict(unet=UNetConfig(), vae=VAEConfig(), pos_encoding=True, cond_channels=0),
    # ldm/configs/nuscenes.yaml:20-27  (1024x32 images -> 256x8 latents)
    "nuscenes": dict(unet=UNetConfig(sample_size=(256, 8)), vae=VAEConfig(sample_size=(1024, 32)),
                     pos_encoding=True, cond_channels=0),
    # ldm/configs/upsample.yaml:14,20 ; ldm/train_conditional.py:232-251  (4 latent + 8 folded-condition channels)
    "upsample": dict(unet=UNetConfig(in_channels=12), vae=VAEConfig(), pos_encoding=False, cond_channels=8),
    # ldm/train_conditional.py:235 (4 latent + 4 masked-latent + 1 mask)
    "inpainting": dict(unet=UNetConfig(in_channels=9), vae=VAEConfig(), pos_encoding=False, cond_channels=5),
    # ldm/configs/RangeDM.yaml:14-21  (pixel space, no VAE)
    "RangeDM": dict(unet=UNetConfig(sample_size=(1024, 64), in_channels=3, out_channels=2,
                                    block_out_channels=(128, 128, 256, 256, 512, 512),
                                    down_block_types=(_D, _D, _D, _D, _A, _D),
                                    up_block_types=(_U, _AU, _U, _U, _U, _U)),
                    vae=None, pos_encoding=True, cond_channels=0),
}
