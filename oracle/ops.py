"""Leaf ops of the oracle (fp32 torch, CPU)."""
import math
import torch
import torch.nn.functional as F


def circ_conv2d(x, weight, bias, stride=1, padding=1):
    """Circular-W / zero-H convolution.  Follows ldm/utils.py:40-55 (dup vae/sgm/.../model.py:93-108):
    pad dim2 (W, azimuth) by wrap-around, pad dim3 (H, beams) by zeros, then conv with padding 0."""
    if padding > 0:
        x = F.pad(x, (0, 0, padding, padding), mode="circular")
        x = F.pad(x, (padding, padding, 0, 0), mode="constant")
    return F.conv2d(x, weight, bias, stride, 0)


def downsample_unet(x, weight, bias):
    """UNet Downsample2D(padding=1, name='op'): circular 3x3 stride 2 pad 1.  ldm/utils.py:92-93,107-116."""
    return circ_conv2d(x, weight, bias, stride=2, padding=1)


def downsample_vae(x, weight, bias):
    """VAE downsample: end-only pad (W: +1 wrap, H: +1 zero) then 3x3 stride 2 pad 0.
    ldm/utils.py:109-111 (padding==0 branch) == vae/sgm/.../model.py:164-172 (circular branch)."""
    x = F.pad(x, (0, 0, 0, 1), mode="circular")
    x = F.pad(x, (0, 1, 0, 0), mode="constant")
    return F.conv2d(x, weight, bias, 2, 0)


def upsample_conv(x, weight, bias):
    """nearest x2 on both dims then circular 3x3.  vae/sgm/.../model.py:120-125; diffusers Upsample2D [3P]."""
    x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    return circ_conv2d(x, weight, bias, 1, 1)


def group_norm_silu(x, w, b, groups, eps, silu=True):
    h = F.group_norm(x, groups, w, b, eps)
    return F.silu(h) if silu else h


def timestep_embedding(t, dim=128, max_period=10000.0, flip_sin_to_cos=True, freq_shift=0):
    """diffusers get_timestep_embedding / Timesteps(dim, flip_sin_to_cos, downscale_freq_shift) [3P; SURVEY.md A.2]:
    f_i = exp(-ln(1e4) * i / (half - freq_shift));  e = [sin(t f), cos(t f)], halves swapped when flip_sin_to_cos.
    UNet2DModel uses (True, 0).  (False, 1) is the reference's own vae/sgm/modules/diffusionmodules/model.py:28-46
    (`[sin, cos]`, divisor `half_dim - 1`), which pins this formula (oracle/validate_unet_against_reference.py)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    s, c = torch.sin(emb), torch.cos(emb)
    return torch.cat([c, s] if flip_sin_to_cos else [s, c], dim=-1)
