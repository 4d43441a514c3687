"""Condition construction for the conditional path (BASELINE config 4: up-sampling 16 -> 64 beams, and in-painting):
what the reference's dataset, training loop and conditional driver do around `LDMUpscalePipelineRange` / the conditional
training step.  Host-side tensor slicing only (device tensors in, device tensors out); the arithmetic that matters (VAE
encode, UNet, sampler) stays in librangeldm_hip.

  ldm/dataset.py:340-346          `down`  = jpg[:, d0//2::d0, d1//2::d1]                       -> downsample_range_image
  ldm/dataset.py:348-362          in-painting mask (+1 inside the masked azimuth span, -1 outside) and the masked image
                                  (-1 where masked)                                            -> inpainting_inputs
  ldm/train_conditional.py:418-430  condition = encoder(down)  |  cat([vae.encode(masked).sample() * sf, resized mask])
                                                                                               -> encode_condition
  ldm/inference_conditional.py:176-182  the -1-filled full-resolution image that shows the sparse input -> sparse_input_image
"""
import torch


def _pair(downsample):
    return [1, downsample] if isinstance(downsample, int) else list(downsample)


def downsample_range_image(range_image, downsample):
    """(..., C, W, H) -> (..., C, W/d0, H/d1): every d-th azimuth column / beam starting at d//2 (ldm/dataset.py:340-346)."""
    d0, d1 = _pair(downsample)
    return range_image[..., (d0 // 2)::d0, (d1 // 2)::d1]


def inpainting_inputs(range_image, fraction, start=0.0):
    """(C, W, H) or (B, C, W, H) -> (inpainting_mask (.., 1, W, H) with +1 over the masked azimuth span [start, start+fraction)
    (wrapping past the seam) and -1 elsewhere, masked_image = range_image outside the span, -1 inside).  ldm/dataset.py:348-362."""
    x = range_image
    W, H = x.shape[-2], x.shape[-1]
    mask = -torch.ones((*x.shape[:-3], 1, W, H), dtype=x.dtype, device=x.device)
    end = start + fraction
    if end < 1.0:
        mask[..., int(start * W):int(end * W), :] = 1
    else:
        mask[..., int(start * W):, :] = 1
        mask[..., :int((end - 1.0) * W), :] = 1
    masked = torch.where(mask < 0, x, -torch.ones_like(x))
    return mask, masked


def encode_condition(batch, vae=None, upsample=True, condition_encoder=None, generator=None, noise=None):
    """The `# encode condition` block of ldm/train_conditional.py:418-430 (and of LDMUpscalePipelineRange.__call__,
    ldm/pipelines.py:473-478 / encode_masked_image :406-412): the (B, Cc, W/4, H/4) tensor concatenated to the noisy latents.
    upsample: condition_encoder(batch["down"]) (SparseRangeImageEncoder2: 8 channels); else cat([VAE latents of
    batch["masked_image"] * scaling_factor, nearest-resized batch["inpainting_mask"]]) (4 + 1 channels)."""
    if upsample:
        if condition_encoder is None:
            from .encoders import SparseRangeImageEncoder2
            condition_encoder = SparseRangeImageEncoder2()
        return condition_encoder(batch["down"])
    if vae is None:
        raise ValueError("the in-painting condition needs the VAE")
    dist = vae.encode(batch["masked_image"]).latent_dist
    lat = dist.sample(noise=noise) if noise is not None else dist.sample(generator=generator)
    lat = lat * vae.config.scaling_factor
    mask = torch.nn.functional.interpolate(batch["inpainting_mask"].to(lat.device), size=lat.shape[-2:])
    return torch.cat([lat, mask.to(lat.dtype)], dim=1)


def sparse_input_image(full_like, down, downsample):
    """The sparse input shown next to the result: -1 everywhere, the low-resolution samples at their source positions
    (ldm/inference_conditional.py:176-182)."""
    d0, d1 = _pair(downsample)
    out = -torch.ones_like(full_like)
    out[..., (d0 // 2)::d0, (d1 // 2)::d1] = down
    return out
