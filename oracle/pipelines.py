"""Oracle denoising loops (fp32 torch).  Follow the four `__call__` bodies of ldm/pipelines.py:
DDPMPipelineRange :34-117, DDIMPipelineRange :144-258, LDMPipelineRange :282-383, LDMUpscalePipelineRange :414-519.
Noise is always injectable (`x_T`, `step_noise[i]`) so parity runs never depend on RNG streams."""
import inspect
import torch


def sparse_range_image_encoder2(x):
    """ldm/encoders.py:90-95: out[b, (w%4)*C + c, w//4, h] = in[b, c, w, h]."""
    B, C, W, H = x.shape
    x = torch.flatten(x.permute(0, 2, 1, 3), start_dim=1, end_dim=2)
    return x.reshape(B, W // 4, C * 4, H).permute(0, 2, 1, 3)


def pos_encoding_channel(B, W, H):
    """ldm/pipelines.py:229-232,346-349: one extra channel, 1 at azimuth index 0."""
    pe = torch.zeros(B, 1, W, H)
    pe[:, :, 0, :] = 1
    return pe


@torch.no_grad()
def ddpm_pipeline(unet, scheduler, x_T, num_inference_steps=1000, step_noise=None, generator=None, trajectory=None):
    """Pixel-space ancestral sampling, no pos-encoding channel.  ldm/pipelines.py:81-108: x_T has the UNet's
    `in_channels`; `scheduler.step(model_output, t, image, generator=generator)` draws one z per step with t > 0."""
    image = x_T.clone()
    scheduler.set_timesteps(num_inference_steps)
    for i, t in enumerate(scheduler.timesteps):
        eps = unet(image, t).sample
        if trajectory is not None:
            trajectory.append((image.clone(), eps.clone()))
        image = scheduler.step(eps, t, image, generator=generator,
                               noise=None if step_noise is None else step_noise[i]).prev_sample
    return image


@torch.no_grad()
def ddim_pipeline(unet, scheduler, x_T, num_inference_steps=50, eta=0.0, pos_encoding=True, step_noise=None,
                  trajectory=None):
    """Pixel-space DDIM (RangeDM).  ldm/pipelines.py:224-248."""
    image = x_T.clone()
    scheduler.set_timesteps(num_inference_steps)
    pe = pos_encoding_channel(*[image.shape[i] for i in (0, 2, 3)]) if pos_encoding else None
    for i, t in enumerate(scheduler.timesteps):
        model_input = torch.cat([image, pe], dim=1) if pos_encoding else image
        eps = unet(model_input, t).sample
        if trajectory is not None:
            trajectory.append((image.clone(), eps.clone()))
        image = scheduler.step(eps, t, image, eta=eta, noise=None if step_noise is None else step_noise[i]).prev_sample
    return image


@torch.no_grad()
def ldm_pipeline(vae, unet, scheduler, x_T, num_inference_steps=50, eta=0.0, pos_encoding=True, step_noise=None,
                 cond=None, trajectory=None, decode=True):
    """Latent sampling + VAE decode.  ldm/pipelines.py:329-367 (unconditional, pos-encoding channel) and :466-507
    (conditional: `cond` (B, C_c, W, H) concatenated every step)."""
    latents = x_T.clone() * scheduler.init_noise_sigma
    scheduler.set_timesteps(num_inference_steps)
    accepts_eta = "eta" in set(inspect.signature(scheduler.step).parameters.keys())
    pe = pos_encoding_channel(*[latents.shape[i] for i in (0, 2, 3)]) if pos_encoding else None
    for i, t in enumerate(scheduler.timesteps):
        x_in = scheduler.scale_model_input(latents, t)
        if pe is not None:
            x_in = torch.cat([x_in, pe], dim=1)
        if cond is not None:
            x_in = torch.cat([x_in, cond], dim=1)
        eps = unet(x_in, t).sample
        if trajectory is not None:
            trajectory.append((latents.clone(), eps.clone()))
        kw = {"eta": eta} if accepts_eta else {}
        kw["noise"] = None if step_noise is None else step_noise[i]
        latents = scheduler.step(eps, t, latents, **kw).prev_sample
    if not decode:
        return latents
    return vae.decode(latents / vae.config.scaling_factor).sample
