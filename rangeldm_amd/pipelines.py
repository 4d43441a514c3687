"""Sampling pipelines with the reference's names and `__call__` signatures (ldm/pipelines.py):
DDPMPipelineRange :34-117, DDIMPipelineRange :144-258, LDMPipelineRange :282-383, LDMUpscalePipelineRange :414-519.

Two execution modes, same results:
  * `fused=True` (default): the whole loop -- conv_in input packing (pos-encoding / condition concat), UNet, scheduler
    step, VAE decode -- runs as HIP graphs inside librangeldm_hip (`rldm_sample`); one host call per batch.
  * `fused=False`: the reference's Python loop verbatim, each `unet(...)` / `scheduler.step(...)` / `vae.decode(...)`
    going through the C ABI individually (this is what "ldm/pipelines.py calls it unchanged" exercises).
"""
import ctypes as C
import inspect

import numpy as np
import torch

from . import _lib
from .schedulers import DDIMSchedulerHIP, DDPMSchedulerHIP, randn_tensor


class ImagePipelineOutput:
    def __init__(self, images):
        self.images = images


class _FusedSampler:
    """Owns one rldm_sampler per (batch, steps, mode, pos_encoding, cond_channels, eta)."""

    def __init__(self):
        self._cache = {}

    def get(self, unet, vae, scheduler, batch, steps, mode, pos_encoding, cond_channels, eta=0.0):
        # keyed by the model objects themselves (kept alive by the cache entry, so an id is never reused for another
        # model); weights reloaded through load_state_dict are picked up by the library: rldm_sample re-plans and
        # re-captures when the model's generation counter moved (include/rangeldm_hip.h, rldm_unet_finalize)
        pred = int(getattr(scheduler, "prediction_code", 0))
        key = (id(unet), id(vae), batch, steps, mode, bool(pos_encoding), cond_channels, float(eta), pred)
        ent = self._cache.get(key)
        if ent is not None:
            return ent[0]
        scheduler.set_timesteps(steps)
        ts = scheduler.timesteps.numpy().astype(np.int64)
        if mode == 0:
            coef = np.asarray([scheduler.coefficients(int(t), eta) for t in ts], dtype=np.float32)
        else:
            coef = np.asarray([scheduler.coefficients(int(t)) for t in ts], dtype=np.float32)
        coef = np.ascontiguousarray(coef)
        ts = np.ascontiguousarray(ts)
        cfg = _lib.SamplerConfigC()
        cfg.batch, cfg.num_steps, cfg.mode = batch, steps, mode
        cfg.pos_encoding = 1 if pos_encoding else 0
        cfg.cond_channels = cond_channels
        cfg.prediction_type = pred
        cfg.coef = coef.ctypes.data_as(C.POINTER(C.c_float))
        cfg.timesteps = ts.ctypes.data_as(C.POINTER(C.c_int64))
        h = C.c_void_p()
        _lib.check(_lib.lib().rldm_sampler_create(unet._h, vae._h if vae is not None else None, C.byref(cfg),
                                                  C.byref(h)), "rldm_sampler_create")
        self._cache[key] = (h, unet, vae)
        return h

    def run(self, h, x_T, step_noise, cond, out, latents_out=None, check=True):
        """One rldm_sample call.  `check` (default): wait for it and raise RuntimeError if the self-check of its persistent
        launches tripped -- the reference's contract is a correct tensor or an exception (ldm/pipelines.py:218-222,463-464);
        the failed call's outputs are NaN-marked on the device either way, and the sampler has then already rebuilt itself as
        one launch per layer, so calling again works.  check=False keeps the call asynchronous (throughput loops that ask
        `status(h)` themselves before they use the images)."""
        def p(t):
            return C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(_lib.lib().rldm_sample(h, p(x_T), p(step_noise), p(cond), p(out), p(latents_out),
                                          _lib.stream_ptr(x_T.device)), "rldm_sample")
        if check:
            self.status(h)

    def status(self, h):
        """rldm_sampler_status: waits for the sampler's last call; raises if its outputs are invalid."""
        _lib.check(_lib.lib().rldm_sampler_status(h), "rldm_sample (self-check of the persistent launches)")

    def status_all(self):
        for ent in self._cache.values():
            self.status(ent[0])

    def __del__(self):
        try:
            for ent in self._cache.values():
                _lib.lib().rldm_sampler_destroy(ent[0])
        except Exception:
            pass


class _PipelineBase:
    def __init__(self):
        self._fused = _FusedSampler()
        self._progress = {}

    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def device(self):
        return self.unet.device

    @property
    def _execution_device(self):
        return self.unet.device

    def to(self, device=None, *a, **k):
        return self

    def save_pretrained(self, output_dir):
        """`pipeline.save_pretrained(args.output_dir)` (ldm/train_unconditional.py:675): unet/, vae/, scheduler/ and
        model_index.json in the layout ldm/inference.py:46-52 reads back."""
        import os
        self.unet.save_pretrained(os.path.join(output_dir, "unet"))
        vae = getattr(self, "vae", None)
        if vae is not None:
            vae.save_pretrained(os.path.join(output_dir, "vae"))
        self.scheduler.save_pretrained(os.path.join(output_dir, "scheduler"))
        import json
        index = {"_class_name": type(self).__name__, "_diffusers_version": "0.21.0",
                 "unet": ["diffusers", "UNet2DModel"], "scheduler": ["diffusers", "DDPMScheduler"]}
        if vae is not None:
            index["vae"] = ["diffusers", "AutoencoderKL"]
        with open(os.path.join(output_dir, "model_index.json"), "w") as f:
            json.dump(index, f, indent=2, sort_keys=True)

    def set_progress_bar_config(self, **kw):
        self._progress = kw

    def progress_bar(self, iterable):
        return iterable

    @staticmethod
    def _finish(image, output_type, return_dict):
        if output_type == "torch":
            return image
        image = (image / 2 + 0.5).clamp(0, 1)
        image = image.cpu().permute(0, 2, 3, 1).numpy()
        if output_type == "pil":
            raise NotImplementedError("output_type='pil' (PIL is not part of the hot path); use 'torch' or 'np'")
        if not return_dict:
            return (image,)
        return ImagePipelineOutput(images=image)

    def _draw_step_noise(self, n_steps, timesteps, shape, generator, device):
        """One randn per step with t > 0, in loop order.  The reference's latent pipelines call `scheduler.step` without a
        generator (ldm/pipelines.py:362,502), i.e. they draw from the unseeded global RNG of the device; here the caller's
        `generator` seeds the step noise as well (after x_T, in loop order), so a seeded call is reproducible.  With
        generator=None the global device RNG is used, as in the reference."""
        zs = torch.zeros((n_steps, *shape), device=device, dtype=torch.float32)
        for i, t in enumerate(timesteps):
            if int(t) > 0:
                zs[i] = randn_tensor(shape, generator=generator, device=device, dtype=torch.float32)
        return zs


class DDPMPipelineRange(_PipelineBase):
    """ldm/pipelines.py:13-117 (pixel-space ancestral sampling, no pos-encoding channel)."""

    def __init__(self, unet, scheduler):
        super().__init__()
        self.register_modules(unet=unet, scheduler=scheduler)

    @torch.no_grad()
    def __call__(self, batch_size=1, generator=None, num_inference_steps=1000, output_type="torch", return_dict=True,
                 fused=True, latents=None, step_noise=None, **kwargs):
        """`latents` / `step_noise` (not in the reference signature): x_T and the [steps][B, C, W, H] ancestral noise
        already on the device, instead of drawing them from `generator`."""
        cfg = self.unet.config
        ss = cfg.sample_size if not isinstance(cfg.sample_size, int) else (cfg.sample_size, cfg.sample_size)
        shape = (batch_size, cfg.in_channels, *ss)
        if latents is not None:
            if tuple(latents.shape) != shape:
                raise ValueError(f"latents shape {tuple(latents.shape)} != {shape}")
            image = latents.to(device=self.device, dtype=torch.float32)
        else:
            image = randn_tensor(shape, generator=generator, device=self.device, dtype=torch.float32)
        self.scheduler.set_timesteps(num_inference_steps)
        if fused and isinstance(self.scheduler, DDPMSchedulerHIP):
            zs = step_noise if step_noise is not None else self._draw_step_noise(
                num_inference_steps, self.scheduler.timesteps, shape, generator, self.device)
            zs = zs.to(self.device, torch.float32).contiguous()
            h = self._fused.get(self.unet, None, self.scheduler, batch_size, num_inference_steps, 1, False, 0)
            out = torch.empty_like(image)
            self._fused.run(h, image.contiguous(), zs, None, out, check=kwargs.get("check", True))
            image = out
        else:
            for i, t in enumerate(self.progress_bar(self.scheduler.timesteps)):
                model_output = self.unet(image, t).sample
                kw = {"noise": step_noise[i]} if step_noise is not None else {"generator": generator}
                image = self.scheduler.step(model_output, t, image, **kw).prev_sample
        return self._finish(image, output_type, return_dict)


class DDIMPipelineRange(_PipelineBase):
    """ldm/pipelines.py:119-258 (RangeDM: pixel-space DDIM with the pos-encoding channel)."""

    def __init__(self, unet, scheduler, pos_encoding=False):
        super().__init__()
        scheduler = DDIMSchedulerHIP.from_config(scheduler.config)      # ldm/pipelines.py:135-139
        self.register_modules(unet=unet, scheduler=scheduler)
        self.pos_encoding = pos_encoding

    @torch.no_grad()
    def __call__(self, batch_size=1, generator=None, eta=0.0, num_inference_steps=50, use_clipped_model_output=None,
                 output_type="torch", return_dict=True, fused=True, latents=None, **kwargs):
        """`latents` (not in the reference signature): x_T already resident on the device, instead of drawing it."""
        cfg = self.unet.config
        ss = cfg.sample_size if not isinstance(cfg.sample_size, int) else (cfg.sample_size, cfg.sample_size)
        shape = (batch_size, cfg.out_channels, *ss)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(
                f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is not None:
            if tuple(latents.shape) != shape:
                raise ValueError(f"latents shape {tuple(latents.shape)} != {shape}")
            image = latents.to(device=self._execution_device, dtype=self.unet.dtype)
        else:
            image = randn_tensor(shape, generator=generator, device=self._execution_device, dtype=self.unet.dtype)
        self.scheduler.set_timesteps(num_inference_steps)
        if fused and eta == 0.0:
            h = self._fused.get(self.unet, None, self.scheduler, batch_size, num_inference_steps, 0,
                                self.pos_encoding, 0, eta)
            out = torch.empty_like(image)
            self._fused.run(h, image.contiguous(), None, None, out, check=kwargs.get("check", True))
            image = out
        else:
            if self.pos_encoding:
                pos_encoding = torch.zeros([shape[0], 1, shape[2], shape[3]], device=self.device)
                pos_encoding[:, :, 0, :] = 1
            for t in self.progress_bar(self.scheduler.timesteps):
                model_input = image
                if self.pos_encoding:
                    model_input = torch.cat([model_input, pos_encoding[:model_input.shape[0]]], dim=1)
                model_output = self.unet(model_input, t).sample
                image = self.scheduler.step(model_output, t, image, eta=eta,
                                            use_clipped_model_output=use_clipped_model_output,
                                            generator=generator).prev_sample
        return self._finish(image, output_type, return_dict)


class LDMPipelineRange(_PipelineBase):
    """ldm/pipelines.py:261-383 (latent sampling + VAE decode).  With a DDPM scheduler this is the strided ancestral
    sampler the reference actually runs; pass a DDIMSchedulerHIP for the BASELINE's DDIM eta=0."""

    def __init__(self, vae, unet, scheduler, pos_encoding=False):
        super().__init__()
        self.register_modules(vae=vae, unet=unet, scheduler=scheduler)
        self.pos_encoding = pos_encoding

    @torch.no_grad()
    def __call__(self, batch_size=1, generator=None, eta=0.0, num_inference_steps=50, output_type="torch",
                 return_dict=True, final_only=True, fused=True, latents=None, step_noise=None, **kwargs):
        cfg = self.unet.config
        shape = (batch_size, cfg.out_channels, *cfg.sample_size)
        if latents is None:
            latents = randn_tensor(shape, generator=generator)          # CPU draw, as ldm/pipelines.py:329-333
        latents = latents.to(self.device)
        latents = latents * self.scheduler.init_noise_sigma
        self.scheduler.set_timesteps(num_inference_steps)
        accepts_eta = "eta" in set(inspect.signature(self.scheduler.step).parameters.keys())
        is_ddim = isinstance(self.scheduler, DDIMSchedulerHIP)
        if fused and final_only and (not is_ddim or eta == 0.0):
            mode = 0 if is_ddim else 1
            zs = None
            if mode == 1:
                zs = step_noise if step_noise is not None else self._draw_step_noise(
                    num_inference_steps, self.scheduler.timesteps, shape, generator, self.device)
                zs = zs.to(self.device, torch.float32).contiguous()
            h = self._fused.get(self.unet, self.vae, self.scheduler, batch_size, num_inference_steps, mode,
                                self.pos_encoding, 0)
            f = self.vae._cfg.downscale
            image = torch.empty((batch_size, self.vae._cfg.out_channels, shape[2] * f, shape[3] * f),
                                device=self.device, dtype=torch.float32)
            self._fused.run(h, latents.float().contiguous(), zs, None, image, check=kwargs.get("check", True))
            return self._finish(image, output_type, return_dict)
        extra_kwargs = {"eta": eta} if accepts_eta else {}
        if self.pos_encoding:
            pos_encoding = torch.zeros([shape[0], 1, shape[2], shape[3]], device=self.device)
            pos_encoding[:, :, 0, :] = 1
        if not final_only:
            assert output_type == "torch"
            image_list = []
        for i, t in enumerate(self.progress_bar(self.scheduler.timesteps)):
            if not final_only:
                image_list.append(self.vae.decode(latents / self.vae.config.scaling_factor).sample)
            latent_model_input = self.scheduler.scale_model_input(latents, t)
            if self.pos_encoding:
                latent_model_input = torch.cat([latent_model_input, pos_encoding], dim=1)
            noise_prediction = self.unet(latent_model_input, t).sample
            kw = dict(extra_kwargs)
            if step_noise is not None and not is_ddim:
                kw["noise"] = step_noise[i]
            latents = self.scheduler.step(noise_prediction, t, latents, **kw).prev_sample
        latents = latents / self.vae.config.scaling_factor
        image = self.vae.decode(latents).sample
        if output_type == "torch" and not final_only:
            image_list.append(image)
            return image_list
        return self._finish(image, output_type, return_dict)


class LDMUpscalePipelineRange(_PipelineBase):
    """ldm/pipelines.py:386-519 (conditional: up-sampling via a condition encoder, in-painting via masked VAE latents)."""

    def __init__(self, vae, unet, scheduler):
        super().__init__()
        self.register_modules(vae=vae, unet=unet, scheduler=scheduler)

    def encode_masked_image(self, image, mask, generator=None, noise=None):
        """ldm/pipelines.py:406-412.  `generator` / `noise` (not in the reference signature) seed or inject the draw of
        `latent_dist.sample()`."""
        image = image.to(self.unet.device)
        dist = self.vae.encode(image).latent_dist
        image = dist.sample(noise=noise) if noise is not None else dist.sample(generator=generator)
        image = image * self.vae.config.scaling_factor
        mask = mask.to(self.unet.device)
        mask = torch.nn.functional.interpolate(mask, size=image.shape[-2:])
        return torch.cat([image, mask], dim=1)

    @torch.no_grad()
    def __call__(self, image, mask=None, condition_encoder=None, batch_size=1, generator=None, eta=0.0,
                 num_inference_steps=50, output_type="torch", return_dict=True, fused=True, latents=None,
                 step_noise=None, encode_noise=None, **kwargs):
        if image is None:
            raise ValueError("`image` input cannot be undefined.")
        cfg = self.unet.config
        shape = (batch_size, cfg.out_channels, *cfg.sample_size)
        if latents is None:
            latents = randn_tensor(shape, generator=generator)
        latents = latents.to(self.unet.device)
        if mask is None:
            assert condition_encoder is not None
            image = condition_encoder(image)
        else:
            image = self.encode_masked_image(image, mask, generator=generator, noise=encode_noise)
        image = image.to(dtype=latents.dtype, device=self.unet.device)
        height, width = image.shape[2:]
        assert cfg.in_channels == cfg.out_channels + image.shape[1]
        assert height == cfg.sample_size[0]
        assert width == cfg.sample_size[1]
        latents = latents * self.scheduler.init_noise_sigma
        self.scheduler.set_timesteps(num_inference_steps)
        accepts_eta = "eta" in set(inspect.signature(self.scheduler.step).parameters.keys())
        is_ddim = isinstance(self.scheduler, DDIMSchedulerHIP)
        if fused and (not is_ddim or eta == 0.0):
            mode = 0 if is_ddim else 1
            zs = None
            if mode == 1:
                zs = step_noise if step_noise is not None else self._draw_step_noise(
                    num_inference_steps, self.scheduler.timesteps, shape, generator, self.device)
                zs = zs.to(self.device, torch.float32).contiguous()
            h = self._fused.get(self.unet, self.vae, self.scheduler, batch_size, num_inference_steps, mode, False,
                                image.shape[1])
            f = self.vae._cfg.downscale
            out = torch.empty((batch_size, self.vae._cfg.out_channels, shape[2] * f, shape[3] * f),
                              device=self.device, dtype=torch.float32)
            self._fused.run(h, latents.float().contiguous(), zs, image.float().contiguous(), out, check=kwargs.get("check", True))
            return self._finish(out, output_type, return_dict)
        extra_kwargs = {"eta": eta} if accepts_eta else {}
        for i, t in enumerate(self.progress_bar(self.scheduler.timesteps)):
            latent_model_input = self.scheduler.scale_model_input(latents, t)
            latent_model_input = torch.cat([latent_model_input, image], dim=1)
            noise_prediction = self.unet(latent_model_input, t).sample
            kw = dict(extra_kwargs)
            if step_noise is not None and not is_ddim:
                kw["noise"] = step_noise[i]
            latents = self.scheduler.step(noise_prediction, t, latents, **kw).prev_sample
        latents = latents / self.vae.config.scaling_factor
        out = self.vae.decode(latents).sample
        return self._finish(out, output_type, return_dict)
