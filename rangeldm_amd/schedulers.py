"""DDPMSchedulerHIP / DDIMSchedulerHIP -- mirror of the diffusers scheduler surface the reference touches
(SURVEY.md 8b, Appendix B): set_timesteps, timesteps, step(...).prev_sample, scale_model_input, init_noise_sigma,
add_noise, alphas_cumprod, config.  Coefficients are computed on the host in fp32 exactly as diffusers does
(torch.linspace / cumprod in float32); the elementwise update runs in librangeldm_hip."""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib
from .config import SchedulerConfig


class SchedulerOutput:
    def __init__(self, prev_sample, pred_original_sample=None):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


def randn_tensor(shape, generator=None, device=None, dtype=None):
    """diffusers.utils.torch_utils.randn_tensor semantics: a CPU generator draws on the CPU and moves."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    rand_device = device
    if generator is not None:
        g0 = generator[0] if isinstance(generator, list) else generator
        if g0.device.type != device.type and g0.device.type == "cpu":
            rand_device = torch.device("cpu")
    if isinstance(generator, list):
        shape1 = (1,) + tuple(shape[1:])
        x = torch.cat([torch.randn(shape1, generator=g, device=rand_device, dtype=dtype) for g in generator], 0)
    else:
        x = torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype)
    return x.to(device)


# scheduler.config.prediction_type -> RLDM_PRED_* (include/rangeldm_hip.h, rldm_sched_step)
PREDICTION_TYPES = {"epsilon": 0, "v_prediction": 1, "sample": 2}


class _SchedulerBase:
    init_noise_sigma = 1.0
    order = 1

    def __init__(self, config=None, **kwargs):
        if config is None:
            config = SchedulerConfig(**kwargs)
        elif isinstance(config, dict):
            config = SchedulerConfig(**{k: v for k, v in config.items() if k in SchedulerConfig.__dataclass_fields__})
        elif not isinstance(config, SchedulerConfig):          # a SimpleNamespace / other scheduler's .config
            config = SchedulerConfig(**{k: getattr(config, k) for k in SchedulerConfig.__dataclass_fields__
                                        if hasattr(config, k)})
        c = self._cfg = config
        if c.beta_schedule != "linear" or c.timestep_spacing != "leading":
            raise NotImplementedError("only the reference's scheduler config is supported (linear betas, leading spacing)")
        if c.prediction_type not in PREDICTION_TYPES:
            # (the message of diffusers' schedulers; ldm/train_unconditional.py:505-510 accepts epsilon and v_prediction)
            raise ValueError(f"prediction_type given as {c.prediction_type} must be one of `epsilon`, `sample` or `v_prediction`")
        self.prediction_code = PREDICTION_TYPES[c.prediction_type]
        if c.clip_sample:
            raise NotImplementedError("clip_sample=True (the reference sets clip_sample=False)")
        self.config = SimpleNamespace(**c.to_dict())
        self.betas = torch.linspace(c.beta_start, c.beta_end, c.num_train_timesteps, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.final_alpha_cumprod = torch.tensor(1.0) if c.set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.arange(c.num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    @classmethod
    def from_config(cls, config, **kw):
        return cls(config, **kw)

    @classmethod
    def load_config(cls, path, subfolder=None):
        """`DDPMScheduler.load_config(args.scheduler_config)` (ldm/inference.py:126): json path or directory."""
        import json
        import os
        from .checkpoint import SCHEDULER_CONFIG_NAME, scheduler_config_from_diffusers
        if subfolder:
            path = os.path.join(path, subfolder)
        if os.path.isdir(path):
            path = os.path.join(path, SCHEDULER_CONFIG_NAME)
        with open(path) as f:
            return scheduler_config_from_diffusers(json.load(f))

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        return cls(cls.load_config(path, subfolder), **kw)

    def save_pretrained(self, path):
        import json
        import os
        from .checkpoint import SCHEDULER_CONFIG_NAME, scheduler_config_to_diffusers
        os.makedirs(path, exist_ok=True)
        name = "DDIMScheduler" if type(self).__name__.startswith("DDIM") else "DDPMScheduler"
        with open(os.path.join(path, SCHEDULER_CONFIG_NAME), "w") as f:
            json.dump(scheduler_config_to_diffusers(self._cfg, name), f, indent=2, sort_keys=True)

    def set_timesteps(self, num_inference_steps, device=None):
        c = self._cfg
        if num_inference_steps > c.num_train_timesteps:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = c.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + c.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _prev_t(self, t):
        n = self.num_inference_steps or self._cfg.num_train_timesteps
        return t - self._cfg.num_train_timesteps // n

    def _alphas(self, t):
        prev_t = self._prev_t(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self._alpha_final()
        return a_t, a_prev

    def add_noise(self, original_samples, noise, timesteps):
        x0 = original_samples.to(dtype=torch.float32).contiguous()
        nz = noise.to(device=x0.device, dtype=torch.float32).contiguous()
        t = timesteps.detach().to("cpu", torch.int64).reshape(-1)
        a = self.alphas_cumprod[t]
        sa = (a ** 0.5).numpy().astype(np.float32)
        sb = ((1 - a) ** 0.5).numpy().astype(np.float32)
        B = x0.shape[0]
        out = torch.empty_like(x0)
        _lib.check(_lib.lib().rldm_sched_add_noise(
            C.c_void_p(x0.data_ptr()), C.c_void_p(nz.data_ptr()), sa.ctypes.data_as(C.POINTER(C.c_float)),
            sb.ctypes.data_as(C.POINTER(C.c_float)), B, x0.numel() // B, C.c_void_p(out.data_ptr()),
            _lib.stream_ptr(x0.device)), "rldm_sched_add_noise")
        return out

    def get_velocity(self, sample, noise, timesteps):
        """`DDPMScheduler.get_velocity` (the v_prediction target, ldm/train_unconditional.py:507-508):
        v = sqrt(alpha_prod_t) * noise - sqrt(1 - alpha_prod_t) * sample.  The same elementwise map as add_noise with the two
        tensors swapped and the second coefficient negated (rldm_sched_add_noise)."""
        x0 = sample.to(dtype=torch.float32).contiguous()
        nz = noise.to(device=x0.device, dtype=torch.float32).contiguous()
        t = timesteps.detach().to("cpu", torch.int64).reshape(-1)
        a = self.alphas_cumprod[t]
        sa = (a ** 0.5).numpy().astype(np.float32)
        sb = (-((1 - a) ** 0.5)).numpy().astype(np.float32)
        B = x0.shape[0]
        out = torch.empty_like(x0)
        _lib.check(_lib.lib().rldm_sched_add_noise(
            C.c_void_p(nz.data_ptr()), C.c_void_p(x0.data_ptr()), sa.ctypes.data_as(C.POINTER(C.c_float)),
            sb.ctypes.data_as(C.POINTER(C.c_float)), B, x0.numel() // B, C.c_void_p(out.data_ptr()),
            _lib.stream_ptr(x0.device)), "rldm_sched_add_noise (get_velocity)")
        return out

    def _launch(self, sampler_mode, coef, model_output, sample, noise):
        e = model_output.to(dtype=torch.float32).contiguous()
        x = sample.to(device=e.device, dtype=torch.float32).contiguous()
        nz = None if noise is None else noise.to(device=e.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(x)
        cf = (C.c_float * 5)(*[float(v) for v in coef])
        _lib.check(_lib.lib().rldm_sched_step(sampler_mode, self.prediction_code, cf, C.c_void_p(e.data_ptr()), C.c_void_p(x.data_ptr()),
                                              C.c_void_p(nz.data_ptr()) if nz is not None else None, C.c_void_p(out.data_ptr()),
                                              x.numel(), _lib.stream_ptr(e.device)), "scheduler step")
        return out


class DDPMSchedulerHIP(_SchedulerBase):
    """Strided ancestral DDPM, variance_type fixed_small (what `LDMPipelineRange` runs as shipped)."""

    def _alpha_final(self):
        return self.one

    def coefficients(self, t):
        t = int(t)
        a_t, a_prev = self._alphas(t)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        sigma = torch.clamp(b_prev / b_t * cur_b, min=1e-20) ** 0.5 if t > 0 else torch.tensor(0.0)
        return [float(a_t ** 0.5), float(b_t ** 0.5), float((a_prev ** 0.5 * cur_b) / b_t),
                float(cur_a ** 0.5 * b_prev / b_t), float(sigma)]

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, noise=None):
        coef = self.coefficients(timestep)
        if coef[4] != 0.0 and noise is None:
            noise = randn_tensor(model_output.shape, generator=generator, device=model_output.device,
                                 dtype=torch.float32)
        prev = self._launch(1, coef, model_output, sample, noise if coef[4] != 0.0 else None)
        return SchedulerOutput(prev) if return_dict else (prev,)


class DDIMSchedulerHIP(_SchedulerBase):
    def _alpha_final(self):
        return self.final_alpha_cumprod

    def coefficients(self, t, eta=0.0):
        t = int(t)
        a_t, a_prev = self._alphas(t)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        var = (b_prev / b_t) * (1 - a_t / a_prev)
        std = eta * var ** 0.5
        return [float(a_t ** 0.5), float(b_t ** 0.5), float(a_prev ** 0.5), float((1 - a_prev - std ** 2) ** 0.5),
                float(std)]

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        coef = self.coefficients(timestep, eta)
        noise = variance_noise
        if coef[4] != 0.0 and noise is None:
            noise = randn_tensor(model_output.shape, generator=generator, device=model_output.device,
                                 dtype=torch.float32)
        prev = self._launch(0, coef, model_output, sample, noise if coef[4] != 0.0 else None)
        return SchedulerOutput(prev) if return_dict else (prev,)
