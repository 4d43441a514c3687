"""Oracle DDPM / DDIM schedulers (fp32 torch).  Restates diffusers `DDPMScheduler` / `DDIMScheduler` [3P, absent from
/root/reference; SURVEY.md Appendix B] with the config the reference builds (ldm/train_unconditional.py:347-352:
1000 linear betas 1e-4..0.02, epsilon prediction, clip_sample=False; DDIM via `from_config`, ldm/pipelines.py:139).
Call sites that fix the semantics: ldm/pipelines.py:99-106,227-246,336-362; ldm/train_unconditional.py:498."""
import torch
from rangeldm_amd.config import SchedulerConfig


class _Out:
    def __init__(self, prev_sample, pred_original_sample=None):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class _Base:
    init_noise_sigma = 1.0

    def __init__(self, config: SchedulerConfig = None):
        self.config = config or SchedulerConfig()
        c = self.config
        assert c.beta_schedule == "linear" and c.timestep_spacing == "leading"
        assert c.prediction_type in ("epsilon", "v_prediction", "sample")
        self.betas = torch.linspace(c.beta_start, c.beta_end, c.num_train_timesteps, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.num_inference_steps = None
        self.timesteps = torch.arange(c.num_train_timesteps - 1, -1, -1, dtype=torch.long)

    def set_timesteps(self, num_inference_steps, device=None):
        c = self.config
        self.num_inference_steps = num_inference_steps
        ratio = c.num_train_timesteps // num_inference_steps
        self.timesteps = (torch.arange(num_inference_steps, dtype=torch.long) * ratio).flip(0) + c.steps_offset

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _prev_t(self, t):
        n = self.num_inference_steps or self.config.num_train_timesteps
        return t - self.config.num_train_timesteps // n

    def _x0_eps(self, model_output, sample, a_t):
        """diffusers' `prediction_type` branches (DDIMScheduler.step / DDPMScheduler.step [3P], published forms):
        (pred_original_sample, pred_epsilon) from the network output."""
        b_t = 1 - a_t
        pt = self.config.prediction_type
        if pt == "epsilon":
            return (sample - b_t ** 0.5 * model_output) / a_t ** 0.5, model_output
        if pt == "sample":
            return model_output, (sample - a_t ** 0.5 * model_output) / b_t ** 0.5
        return a_t ** 0.5 * sample - b_t ** 0.5 * model_output, a_t ** 0.5 * model_output + b_t ** 0.5 * sample

    def get_velocity(self, sample, noise, timesteps):
        """DDPMScheduler.get_velocity [3P]: v = sqrt(alpha_prod) * noise - sqrt(1 - alpha_prod) * sample
        (call site: ldm/train_unconditional.py:508)."""
        a = self.alphas_cumprod[timesteps]
        sa = (a ** 0.5).view(-1, *([1] * (sample.dim() - 1)))
        sb = ((1 - a) ** 0.5).view(-1, *([1] * (sample.dim() - 1)))
        return sa * noise - sb * sample

    def add_noise(self, x0, noise, timesteps):
        a = self.alphas_cumprod[timesteps]
        sa = (a ** 0.5).view(-1, *([1] * (x0.dim() - 1)))
        sb = ((1 - a) ** 0.5).view(-1, *([1] * (x0.dim() - 1)))
        return sa * x0 + sb * noise


class OracleDDPMScheduler(_Base):
    def step(self, model_output, timestep, sample, generator=None, noise=None, return_dict=True):
        """Strided ancestral DDPM, variance_type fixed_small (B.3).  `noise` injects z instead of drawing it."""
        t = int(timestep)
        prev_t = self._prev_t(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        x0, _ = self._x0_eps(model_output, sample, a_t)
        mean = (a_prev ** 0.5 * cur_b) / b_t * x0 + cur_a ** 0.5 * b_prev / b_t * sample
        if t > 0:
            if noise is None:
                noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            var = torch.clamp(b_prev / b_t * cur_b, min=1e-20)
            mean = mean + var ** 0.5 * noise
        return _Out(mean, x0)

    def coefficients(self, t):
        """(c_x0, c_xt, sigma) of x_prev = c_x0*x0 + c_xt*x_t + sigma*z, as python floats (for known-answer tests)."""
        prev_t = self._prev_t(int(t))
        a_t = self.alphas_cumprod[int(t)]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        sigma = torch.clamp(b_prev / b_t * cur_b, min=1e-20) ** 0.5 if int(t) > 0 else torch.tensor(0.0)
        return float((a_prev ** 0.5 * cur_b) / b_t), float(cur_a ** 0.5 * b_prev / b_t), float(sigma)


class OracleDDIMScheduler(_Base):
    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=None, generator=None,
             noise=None, return_dict=True):
        """DDIM (B.2); eta>0 adds sigma*z."""
        t = int(timestep)
        prev_t = self._prev_t(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        x0, pred_eps = self._x0_eps(model_output, sample, a_t)
        var = (b_prev / b_t) * (1 - a_t / a_prev)
        std = eta * var ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * pred_eps
        prev = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            if noise is None:
                noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev = prev + std * noise
        return _Out(prev, x0)
