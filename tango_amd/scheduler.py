"""Host-side scheduler logic: mirrors the interface `AudioDiffusion.inference` touches on the
reference scheduler object (`set_timesteps`, `timesteps`, `order`, `init_noise_sigma`,
`scale_model_input`, `step(...).prev_sample`, `config`) -- reference:
mustango/diffusers/src/diffusers/schedulers/scheduling_ddpm.py:123-349 and scheduling_ddim.py:120-360.

The tensor math of `step` runs inside the engine's fused CFG+step kernel; this module only produces
the integer timestep schedule (bit-exact requirement) and the per-step fp32 coefficient table the
kernel consumes.  Tables are computed with torch fp32 CPU ops in the reference's own order
(`linspace(sqrt(b0), sqrt(b1), T)**2`, `cumprod`) so the scalars are bit-identical to the reference's.
"""
from types import SimpleNamespace

import numpy as np
import torch

#: stabilityai/stable-diffusion-2-1 `scheduler/scheduler_config.json` (tango.py:36 fetches it from the
#: hub; not in the tree).  The engine takes the config as data -- pass a different dict to override.
SD21_SCHEDULER_CONFIG = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                             beta_schedule="scaled_linear", prediction_type="v_prediction",
                             clip_sample=False, variance_type="fixed_small", clip_sample_range=1.0,
                             set_alpha_to_one=False, steps_offset=1)


class _StepOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class _SchedulerBase:
    order = 1
    rule = "ddpm"

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 prediction_type="epsilon", clip_sample=True, clip_sample_range=1.0, **extra):
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError("%s does is not implemented for %s" % (beta_schedule, type(self).__name__))
        if prediction_type not in ("epsilon", "sample", "v_prediction"):
            raise ValueError("prediction_type given as %s must be one of `epsilon`, `sample` or `v_prediction`" % prediction_type)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type=prediction_type,
                                      clip_sample=clip_sample, clip_sample_range=clip_sample_range, **extra)

    @classmethod
    def from_config(cls, config: dict):
        return cls(**config)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _base_timesteps(self, n):
        T = self.config.num_train_timesteps
        if n > T:
            raise ValueError(
                "`num_inference_steps`: %d cannot be larger than `self.config.train_timesteps`: %d as the unet model "
                "trained with this scheduler can only handle maximal %d timesteps." % (n, T, T))
        self.num_inference_steps = n
        ratio = T // n
        return (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)

    def _prev(self, t):
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return t - self.config.num_train_timesteps // n


class DDPMScheduler(_SchedulerBase):
    """Stochastic DDPM sampler (what tango.py uses): scheduling_ddpm.py."""
    rule = "ddpm"

    def __init__(self, variance_type="fixed_small", **kw):
        super().__init__(variance_type=variance_type, **kw)
        self.variance_type = variance_type

    def set_timesteps(self, num_inference_steps, device=None):
        self.timesteps = torch.from_numpy(self._base_timesteps(num_inference_steps))

    def _get_variance(self, t):
        prev_t = self._prev(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        cur_beta = 1 - a_t / a_prev
        var = (1 - a_prev) / (1 - a_t) * cur_beta
        if self.variance_type == "fixed_small":
            var = torch.clamp(var, min=1e-20)
        elif self.variance_type == "fixed_large":
            var = cur_beta
        else:
            raise NotImplementedError("variance_type %s" % self.variance_type)
        return var

    def coef_table(self) -> np.ndarray:
        """[N, 8] fp32: sqrt(abar_t), sqrt(1-abar_t), coef_x0, coef_xt, sigma (0 at t == 0), 0, 0, 0"""
        rows = []
        for t in self.timesteps.tolist():
            prev_t = self._prev(t)
            a_t = self.alphas_cumprod[t]
            a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
            b_t, b_prev = 1 - a_t, 1 - a_prev
            cur_a = a_t / a_prev
            cur_b = 1 - cur_a
            c_x0 = (a_prev ** 0.5 * cur_b) / b_t
            c_xt = cur_a ** 0.5 * b_prev / b_t
            sig = self._get_variance(t) ** 0.5 if t > 0 else torch.tensor(0.0)
            rows.append([float(a_t ** 0.5), float(b_t ** 0.5), float(c_x0), float(c_xt), float(sig), 0.0, 0.0, 0.0])
        return np.asarray(rows, dtype=np.float32)


class DDIMScheduler(_SchedulerBase):
    """DDIM rule (fork schedulers/scheduling_ddim.py:238-360; AudioLDM's DDIMSampler.p_sample_ddim, audioldm/latent_diffusion/
    ddim.py:306-377, is the same update): the north-star's optional sampler.  `eta` = 0 (default) is the deterministic rule;
    eta > 0 adds sigma_t = eta * sqrt((1 - abar_prev) / (1 - abar_t) * (1 - abar_t / abar_prev)) of fresh noise per step and
    shortens the direction term to sqrt(1 - abar_prev - sigma_t^2) (scheduling_ddim.py:316-352, ddim.py:356-370; eta = 1 is
    DDPM-like ancestral sampling).  The noise comes from the engine's step-noise source (injected tensor or device Philox),
    like the DDPM rule's."""
    rule = "ddim"

    def __init__(self, set_alpha_to_one=True, steps_offset=0, eta=0.0, **kw):
        super().__init__(set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, **kw)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        if eta < 0:
            raise ValueError("eta must be >= 0")
        self.eta = float(eta)

    def set_timesteps(self, num_inference_steps, device=None):
        self.timesteps = torch.from_numpy(self._base_timesteps(num_inference_steps)) + self.config.steps_offset

    def coef_table(self, eta=None) -> np.ndarray:
        """per-step coefficients of the fused update; `eta` overrides the constructor's value for this table (the fork passes eta to
        every step() call, scheduling_ddim.py:238: here the loop is one engine call, so it is a per-table argument)"""
        eta = self.eta if eta is None else float(eta)
        if eta < 0:
            raise ValueError("eta must be >= 0")
        rows = []
        T = self.config.num_train_timesteps
        for t in self.timesteps.tolist():
            prev_t = t - T // self.num_inference_steps
            a_t = self.alphas_cumprod[t]
            a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
            b_t = 1 - a_t
            var = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)          # scheduling_ddim.py:184-193
            std = eta * var ** 0.5                                           # :316-317 (eta = 0 -> std_dev_t = 0)
            direction = (1 - a_prev - std ** 2) ** 0.5                       # :340
            rows.append([float(a_t ** 0.5), float(b_t ** 0.5), 0.0, 0.0, float(std), float(a_prev ** 0.5), float(direction), 0.0])
        return np.asarray(rows, dtype=np.float32)
