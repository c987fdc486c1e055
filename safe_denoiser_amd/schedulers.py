"""Scheduler front-ends: the diffusers call surface the reference's pipelines use, arithmetic in libsdn.

Surface mirrored (call sites: models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:
489,520,554,569,576; run_nudity.py:108,309-312; repellency/repellency_methods_threshold.py:121,138;
models/sdv3/safe_denoiser_pipeline.py:1085,1103,1165):
    set_timesteps(n, device=...), timesteps, scale_model_input(x, t), step(model_output, t, sample,
    generator=...) -> .prev_sample / .pred_original_sample, add_noise(x0, noise, t), attributes
    betas, alphas_cumprod, init_noise_sigma, order, config.{num_train_timesteps, beta_start, beta_end}.

Per-step coefficient tables (1000 floats) are init-time host work and are built with torch CPU ops in fp32
exactly as diffusers 0.29.0 builds them; the per-element step math (the hot path) is one HIP kernel per call.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from . import _lib


class SchedulerOutput:
    __slots__ = ("prev_sample", "pred_original_sample")

    def __init__(self, prev_sample, pred_original_sample=None):
        self.prev_sample, self.pred_original_sample = prev_sample, pred_original_sample

    def __getitem__(self, i):                       # return_dict=False style access
        return (self.prev_sample,)[i]


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    _lib.require_gpu()
    if t.dtype != torch.float32:
        raise _lib.SdnError(f"{name}: scheduler kernels compute in fp32, got {t.dtype}")
    return t.contiguous()


class _DiscreteScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 steps_offset=1, clip_sample=False, clip_sample_range=1.0, timestep_spacing="leading",
                 prediction_type="epsilon", **unused):
        if beta_schedule != "scaled_linear" or timestep_spacing != "leading" or prediction_type != "epsilon":
            raise NotImplementedError("only the SD-v1.4 scheduler configuration is implemented "
                                      "(scaled_linear / leading / epsilon)")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule, steps_offset=steps_offset,
                                      clip_sample=clip_sample, clip_sample_range=clip_sample_range,
                                      timestep_spacing=timestep_spacing, prediction_type=prediction_type)
        self.beta_start, self.beta_end = beta_start, beta_end
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.num_inference_steps = None
        self.timesteps = None
        self._stride = None

    # -- construction from a checkpoint's scheduler_config.json (run_nudity.py:108) ------------------
    # diffusers' from_config semantics: keys of the file that the class accepts are used, the others are ignored, and a key
    # the file LACKS takes the class's own default -- which for clip_sample is True in DDPM / DDIM (the
    # SD-v1.4 file is PNDM-authored; whether pred_original_sample is clipped to [-1, 1] is decided by that file, not here).
    _CLASS_DEFAULTS = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", steps_offset=0,
                           clip_sample=True, clip_sample_range=1.0, timestep_spacing="leading", prediction_type="epsilon")
    _ACCEPTED = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "steps_offset", "clip_sample",
                 "clip_sample_range", "timestep_spacing", "prediction_type", "set_alpha_to_one")

    @classmethod
    def from_config(cls, config: dict):
        kw = dict(cls._CLASS_DEFAULTS)
        kw.update({k: v for k, v in config.items() if k in cls._ACCEPTED})
        if config.get("trained_betas") is not None:
            raise NotImplementedError("trained_betas is not implemented")
        for k, want in (("variance_type", "fixed_small"), ("thresholding", False), ("rescale_betas_zero_snr", False)):
            if k in config and config[k] != want:
                raise NotImplementedError(f"scheduler config {k} = {config[k]!r} is not implemented")
        return cls(**kw)

    @classmethod
    def from_pretrained(cls, model_dir: str, subfolder: str = "scheduler"):
        from .checkpoint import read_config
        import os
        return cls.from_config(read_config(os.path.join(model_dir, subfolder) if subfolder else model_dir, "scheduler_config.json"))

    # -- tables -----------------------------------------------------------------------------------
    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        if num_inference_steps > T:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        self._stride = T // num_inference_steps
        ts = (torch.arange(0, num_inference_steps, dtype=torch.float64) * self._stride).round().flip(0).to(torch.int64)
        ts = ts + self.config.steps_offset
        self.timesteps = ts.to(device) if device is not None else ts

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _acp(self, t: int) -> torch.Tensor:
        return self.alphas_cumprod[int(t)]

    def _acp_prev(self, t: int) -> torch.Tensor:
        raise NotImplementedError

    def sqrt_pair(self, t) -> tuple[float, float]:
        a = self._acp(t)
        return float(a ** 0.5), float((1 - a) ** 0.5)

    # -- elementwise hot path ------------------------------------------------------------------------
    def add_noise(self, original_samples, noise, timesteps):
        x0, z = _f32(original_samples, "original_samples"), _f32(noise, "noise")
        sa, s1 = self.sqrt_pair(int(timesteps))
        out = torch.empty_like(x0)
        _lib.check(_lib.lib().sdn_add_noise(_lib.dptr(x0), _lib.dptr(z), x0.numel(), sa, s1, _lib.dptr(out),
                                            _lib.stream_ptr()), "sdn_add_noise")
        return out

    def pred_original_sample(self, model_output, timestep, sample):
        e, x = _f32(model_output, "model_output"), _f32(sample, "sample")
        sa, s1 = self.sqrt_pair(int(timestep))
        x0 = torch.empty_like(x)
        clip = self.config.clip_sample_range if self.config.clip_sample else 0.0
        _lib.check(_lib.lib().sdn_pred_x0(_lib.dptr(x), _lib.dptr(e), x.numel(), sa, s1, clip, _lib.dptr(x0),
                                          _lib.stream_ptr()), "sdn_pred_x0")
        return x0

    def step_coefficients(self, t: int) -> dict:
        raise NotImplementedError

    def _launch_step(self, e, x, z, co, out):
        clip = self.config.clip_sample_range if self.config.clip_sample else 0.0
        _lib.check(_lib.lib().sdn_sched_step(_lib.dptr(x), _lib.dptr(e), _lib.dptr(z), x.numel(), co["sqrt_ac"],
                                             co["sqrt_1mac"], co["c_x0"], co["c_x"], co["c_eps"], co["sigma"], clip,
                                             _lib.dptr(out), _lib.stream_ptr()), "sdn_sched_step")


class DDPMScheduler(_DiscreteScheduler):
    """Ancestral DDPM, variance_type fixed_small -- the live SD-v1.4 scheduler (run_nudity.py:108)."""

    def step_coefficients(self, t: int) -> dict:
        t = int(t)
        pt = t - self._stride
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[pt] if pt >= 0 else torch.tensor(1.0)
        b_t, b_p = 1 - a_t, 1 - a_p
        cur_alpha = a_t / a_p
        cur_beta = 1 - cur_alpha
        var = torch.clamp(b_p / b_t * cur_beta, min=1e-20)
        return {"sqrt_ac": float(a_t ** 0.5), "sqrt_1mac": float(b_t ** 0.5),
                "c_x0": float(a_p ** 0.5 * cur_beta / b_t), "c_x": float(cur_alpha ** 0.5 * b_p / b_t), "c_eps": 0.0,
                "sigma": float(var ** 0.5) if t > 0 else 0.0}

    def step(self, model_output, timestep, sample, generator=None, noise=None, return_dict=True,
             want_pred_original_sample=True):
        """One reverse step.  Draws the variance noise from `generator` exactly where diffusers does (one
        randn of model_output.shape per call for t > 0) unless `noise` is supplied by the caller."""
        e, x = _f32(model_output, "model_output"), _f32(sample, "sample")
        t = int(timestep)
        co = self.step_coefficients(t)
        z = None
        if t > 0:
            z = noise if noise is not None else torch.randn(e.shape, generator=generator, device=e.device,
                                                            dtype=e.dtype)
            z = _f32(z, "noise")
        prev = torch.empty_like(x)
        self._launch_step(e, x, z, co, prev)
        x0 = self.pred_original_sample(e, t, x) if want_pred_original_sample else None
        return SchedulerOutput(prev, x0)


class DDIMScheduler(_DiscreteScheduler):
    """DDIM, eta = 0, set_alpha_to_one = False (BASELINE.json names it; commented out at run_nudity.py:107)."""

    _CLASS_DEFAULTS = dict(_DiscreteScheduler._CLASS_DEFAULTS, set_alpha_to_one=True)

    def __init__(self, *a, set_alpha_to_one=False, **kw):
        super().__init__(*a, **kw)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def step_coefficients(self, t: int) -> dict:
        t = int(t)
        pt = t - self._stride
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[pt] if pt >= 0 else self.final_alpha_cumprod
        return {"sqrt_ac": float(a_t ** 0.5), "sqrt_1mac": float((1 - a_t) ** 0.5), "c_x0": float(a_p ** 0.5),
                "c_x": 0.0, "c_eps": float((1 - a_p) ** 0.5), "sigma": 0.0}

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, return_dict=True,
             want_pred_original_sample=True, **unused):
        if eta != 0.0:
            raise NotImplementedError("DDIM eta > 0 is not on the reference's path")
        e, x = _f32(model_output, "model_output"), _f32(sample, "sample")
        co = self.step_coefficients(int(timestep))
        prev = torch.empty_like(x)
        self._launch_step(e, x, None, co, prev)
        x0 = self.pred_original_sample(e, int(timestep), x) if want_pred_original_sample else None
        return SchedulerOutput(prev, x0)


class FlowMatchEulerDiscreteScheduler:
    """diffusers-0.29.0 flow-matching Euler scheduler, shift 3 for SD-v3
    (models/sdv3/safe_denoiser_pipeline.py:31,256,1085,1103,1165)."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, shift=3.0):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift)
        ts = torch.linspace(1, num_train_timesteps, num_train_timesteps, dtype=torch.float32).flip(0)
        sig = ts / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigma_min, self.sigma_max = float(sig[-1]), float(sig[0])
        self.timesteps, self.sigmas, self._step_index = sig * num_train_timesteps, None, None

    def set_timesteps(self, num_inference_steps: int, device=None):
        T, sh = self.config.num_train_timesteps, self.config.shift
        ts = torch.linspace(self.sigma_max * T, self.sigma_min * T, num_inference_steps, dtype=torch.float32)
        sig = ts / T
        sig = sh * sig / (1 + (sh - 1) * sig)
        self.num_inference_steps = num_inference_steps
        self._sig_host = torch.cat([sig, torch.zeros(1)])
        self.sigmas = self._sig_host.to(device) if device is not None else self._sig_host
        self._ts_host = sig * T
        self.timesteps = self._ts_host.to(device) if device is not None else self._ts_host
        self._step_index = 0

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict=True, **unused):
        _lib.require_gpu()
        v = model_output.float().contiguous()
        x = sample.float().contiguous()
        s, sn = float(self._sig_host[self._step_index]), float(self._sig_host[self._step_index + 1])
        prev = torch.empty_like(x)
        _lib.check(_lib.lib().sdn_flow_euler_step(_lib.dptr(x), _lib.dptr(v), x.numel(), s, sn, _lib.dptr(prev),
                                                  _lib.stream_ptr()), "sdn_flow_euler_step")
        self._step_index += 1
        return SchedulerOutput(prev.to(model_output.dtype))
