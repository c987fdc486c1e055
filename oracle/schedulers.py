"""CPU oracle: scheduler arithmetic (SURVEY.md section 8a rows S1, S1', S2, S3).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

PARITY UNPINNED at the reference level: the arithmetic lives in diffusers==0.29.0
(requirements.txt:3), which is neither under /root/reference nor installed here, and the
reference holds no tests/golden vectors for it.  This file restates the published
diffusers-0.29.0 definitions of DDPMScheduler / DDIMScheduler / FlowMatchEulerDiscreteScheduler
for the configuration the reference loads (SD-v1.4 scheduler json: scaled_linear betas
0.00085..0.012, 1000 train steps, steps_offset 1, clip_sample false, leading spacing,
epsilon prediction, fixed_small variance; SD-v3: shift 3.0), anchored on the reference's call
sites:  run_nudity.py:108,309-312;
        models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:489,520,554,569,576;
        repellency/repellency_methods_threshold.py:121,138;
        models/sdv3/safe_denoiser_pipeline.py:1085,1103,1165.
It is checked by self-consistency known-answer tests (tests/test_oracle_schedulers.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


def scaled_linear_betas(beta_start=0.00085, beta_end=0.012, n=1000) -> torch.Tensor:
    return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2


def leading_timesteps(num_inference_steps: int, num_train=1000, steps_offset=1) -> torch.Tensor:
    """t_i = (i * (T // n))[::-1] + offset  -> 981, 961, ..., 1 for n=50."""
    ratio = num_train // num_inference_steps
    ts = torch.arange(0, num_inference_steps, dtype=torch.float64) * ratio
    return ts.round().flip(0).to(torch.int64) + steps_offset


@dataclass
class StepOut:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor


class _DiscreteBase:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000, steps_offset=1,
                 clip_sample=False, clip_sample_range=1.0):
        self.beta_start, self.beta_end = beta_start, beta_end
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.clip_sample, self.clip_sample_range = clip_sample, clip_sample_range
        self.betas = scaled_linear_betas(beta_start, beta_end, num_train_timesteps)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n: int, device=None):
        self.num_inference_steps = n
        self.timesteps = leading_timesteps(n, self.num_train_timesteps, self.steps_offset)

    def scale_model_input(self, sample, t=None):
        return sample

    def add_noise(self, original, noise, t):
        """sqrt(acp_t) * x0 + sqrt(1 - acp_t) * noise  (...threshold_time.py:569)."""
        acp = self.alphas_cumprod[int(t)]
        return acp ** 0.5 * original + (1 - acp) ** 0.5 * noise

    def _prev_t(self, t: int) -> int:
        return t - self.num_train_timesteps // self.num_inference_steps

    def _x0(self, eps, t, sample):
        acp = self.alphas_cumprod[int(t)]
        x0 = (sample - (1 - acp) ** 0.5 * eps) / acp ** 0.5
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        return x0


class DDPM(_DiscreteBase):
    """Ancestral sampler, variance_type fixed_small (the LIVE SD-v1.4 scheduler, run_nudity.py:108)."""

    def variance(self, t: int) -> torch.Tensor:
        pt = self._prev_t(t)
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[pt] if pt >= 0 else self.one
        cur_beta = 1 - a_t / a_p
        return torch.clamp((1 - a_p) / (1 - a_t) * cur_beta, min=1e-20)

    def step(self, eps, t, sample, generator=None) -> StepOut:
        t = int(t)
        pt = self._prev_t(t)
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[pt] if pt >= 0 else self.one
        b_t, b_p = 1 - a_t, 1 - a_p
        cur_alpha = a_t / a_p
        cur_beta = 1 - cur_alpha
        x0 = self._x0(eps, t, sample)
        c_x0 = (a_p ** 0.5 * cur_beta) / b_t
        c_x = cur_alpha ** 0.5 * b_p / b_t
        prev = c_x0 * x0 + c_x * sample
        if t > 0:                                  # one randn draw per call, even when its weight is 1e-10
            z = torch.randn(eps.shape, generator=generator, dtype=eps.dtype)
            prev = prev + self.variance(t) ** 0.5 * z
        return StepOut(prev, x0)


class DDIM(_DiscreteBase):
    """eta = 0, set_alpha_to_one = False (named by BASELINE.json; commented at run_nudity.py:107)."""

    def step(self, eps, t, sample, generator=None, eta: float = 0.0) -> StepOut:
        t = int(t)
        pt = self._prev_t(t)
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[pt] if pt >= 0 else self.alphas_cumprod[0]
        x0 = self._x0(eps, t, sample)
        var = ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)
        std = eta * var ** 0.5
        direction = (1 - a_p - std ** 2) ** 0.5 * eps
        prev = a_p ** 0.5 * x0 + direction
        if eta > 0:
            prev = prev + std * torch.randn(eps.shape, generator=generator, dtype=eps.dtype)
        return StepOut(prev, x0)


class FlowMatchEuler:
    """FlowMatchEulerDiscreteScheduler, diffusers 0.29.0, shift = 3 (SD-v3).

    0.29.0 applies the shift to the training grid in __init__ (sigma_min = shifted 1/1000) and AGAIN to
    the linspace in set_timesteps -- restated from the published source; unverifiable here.
    """
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, shift=3.0):
        self.num_train_timesteps, self.shift = num_train_timesteps, shift
        ts = torch.linspace(1, num_train_timesteps, num_train_timesteps, dtype=torch.float32).flip(0)
        sig = ts / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigma_min, self.sigma_max = float(sig[-1]), float(sig[0])
        self.timesteps, self.sigmas = sig * num_train_timesteps, None
        self._i = None

    def set_timesteps(self, n: int, device=None):
        T = self.num_train_timesteps
        ts = torch.linspace(self.sigma_max * T, self.sigma_min * T, n, dtype=torch.float32)
        sig = ts / T
        sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        self.timesteps = sig * T
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        self._i = 0

    def step(self, v, t, sample):
        """prev = x + (sigma_next - sigma) * v, fp32 upcast, cast back to v.dtype (safe_denoiser_pipeline.py:1165)."""
        x = sample.float()
        s, sn = self.sigmas[self._i], self.sigmas[self._i + 1]
        denoised = x - v.float() * s
        derivative = (x - denoised) / s
        prev = x + derivative * (sn - s)
        self._i += 1
        return prev.to(v.dtype)


def flow_repellency_renoise(latents, v, sigma, sigma_next, x0_repelled_fn, z):
    """SD-v3 repellency window body, models/sdv3/safe_denoiser_pipeline.py:1139-1161.

    x0 = x - sigma v ; x1 = x + (1-sigma) v ; delta = sigma - sigma_next ;
    noise = sqrt(sigma_next) x1 + sqrt(1-sigma_next) z ; out = x0r + (sigma - delta)(noise - x0r).
    """
    x0 = latents - sigma * v
    x1 = latents + (1 - sigma) * v
    delta = sigma - sigma_next
    x0r = x0_repelled_fn(x0)
    noise = math.sqrt(sigma_next) * x1 + math.sqrt(1 - sigma_next) * z
    return x0r + (sigma - delta) * (noise - x0r)
