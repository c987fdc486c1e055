"""CPU oracle: the reference's denoising loop, one prompt at a time.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Restates
models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:511-582 (and the `_time` / plain
variants' repellency blocks; vanilla loop: models/vanilla/stable_diffusion_pipeline.py
:264-289) on top of oracle.unet / oracle.schedulers / oracle.repellency.  Parity status: the repellency block is
pinned by golden vectors; UNet + scheduler arithmetic is parity-unpinned at the reference level (diffusers absent).

Random numbers come from `noise_fn(prompt_index, shape)` so the CPU oracle and the GPU engine can be fed the SAME
per-prompt noise tape; what this loop pins is the DRAW ORDER of row S2:
    latents -> per step { [window & processor: the x0 probe's scheduler.step draws and discards one randn]
                          -> [is_negation: re-noise randn] -> the step's variance randn }.
"""
from __future__ import annotations

import torch

from . import repellency as orp
from . import schedulers as osch

VARIANTS = {
    "threshold_time": ("t", 780, 1000, True, True),      # {safree,sld}_*_threshold_time.py:429-431,552
    "time": ("t", 800, 1000, False, False),              # *_time.py:430-431,549 (SLD file: lower bound 780, :380-382)
    "threshold": ("i", 0, 50, True, True),               # *_threshold.py:430-431,548 -- hard-coded, kwargs never read
    "sd_threshold_time": ("i", 0, 11, True, True),       # modified_stable_diffusion_pipeline_threshold_time.py:430-431,551
    "plain": (None, 0, 0, False, True),
}


class TapeGenerator:
    """Stands in for torch.Generator: hands out the p-th prompt's pre-generated noise in draw order."""

    def __init__(self, noise_fn, p):
        self.noise_fn, self.p = noise_fn, p

    def randn(self, shape):
        return self.noise_fn(self.p, tuple(shape))


def denoise_one(unet, scheduler, text_pair, p, noise_fn, *, num_inference_steps=50, guidance_scale=7.5,
                repel=None, variant="threshold_time", lra=False, text_safe=None, use_safe_fn=None,
                negation_warmup_start=None, negation_warmup_end=None, sld=None):
    """text_pair: [2,77,768] (uncond, text) for prompt p.  repel: dict(flavour=..., proj_refs=..., **params) or None.
    Returns (final latents [1,C,S,S], stats)."""
    kind, lo_d, hi_d, use_beta, use_flag = VARIANTS[variant]
    if variant == "threshold":
        lo, hi = lo_d, hi_d
    elif kind == "i":                                      # i >= start and i <= end
        lo = lo_d if negation_warmup_start is None else negation_warmup_start
        hi = hi_d if negation_warmup_end is None else negation_warmup_end
    else:                                                  # t <= start and t >= end
        hi = hi_d if negation_warmup_start is None else negation_warmup_start
        lo = (780 if (sld and variant == "time") else lo_d) if negation_warmup_end is None else negation_warmup_end
    gen = TapeGenerator(noise_fn, p)
    is_ddpm = isinstance(scheduler, osch.DDPM)
    scheduler.set_timesteps(num_inference_steps)
    cfg = unet.cfg
    shape = (1, cfg["in_channels"], cfg["sample_size"], cfg["sample_size"])
    latents = gen.randn(shape) * scheduler.init_noise_sigma
    nb = 3 if (lra or sld) else 2
    n_renoise = 0
    momentum = None            # SLD eq. 8 state (modified_sld_pipeline_threshold_time.py:455,474-503)
    for i, t in enumerate(scheduler.timesteps.tolist()):
        x_in = torch.cat([latents] * nb)
        E = text_safe if (text_safe is not None and use_safe_fn is not None and use_safe_fn(i)) else text_pair
        if lra:
            E = torch.cat([E, text_pair[1:2]])
        out = unet(x_in, float(t), E)
        e_u, e_t = out[0:1], out[1:2]
        guide = e_t - e_u
        if sld:                                        # text_pair is then [3,77,768]: uncond, text, safety concept
            e_c = out[2:3]
            if momentum is None:
                momentum = torch.zeros_like(guide)
            scale = torch.clamp(torch.abs(e_t - e_c) * sld["scale"], max=1.0)
            scale = torch.where((e_t - e_c) >= sld["thr"], torch.zeros_like(scale), scale)
            gs = (e_c - e_u) * scale + sld["ms"] * momentum
            momentum = sld["mb"] * momentum + (1 - sld["mb"]) * gs
            if i >= sld["warmup"]:
                guide = guide - gs
        eps = e_u + guidance_scale * guide
        in_window = (kind is None) or (kind == "t" and lo <= t <= hi) or (kind == "i" and lo <= i <= hi)
        if in_window and repel is not None:
            x0_hat = _step(scheduler, eps, t, latents, gen, is_ddpm).pred_original_sample
            d = _conditioning(repel, x0_hat, use_beta)
            if d.get("is_negation", False) if use_flag else True:
                noise = gen.randn(d["x_0_hat"].shape)
                latents = scheduler.add_noise(d["x_0_hat"], noise, t)
                n_renoise += 1
        latents = _step(scheduler, eps, t, latents, gen, is_ddpm).prev_sample
    return latents, {"renoise_draws": n_renoise}


def _step(scheduler, eps, t, latents, gen, is_ddpm):
    if is_ddpm:
        z = gen.randn(eps.shape) if t > 0 else None
        return _ddpm_step_with_noise(scheduler, eps, t, latents, z)
    return scheduler.step(eps, t, latents)


def _ddpm_step_with_noise(s, eps, t, sample, z):
    """oracle.schedulers.DDPM.step with the variance noise supplied by the tape instead of a torch.Generator."""
    t = int(t)
    pt = s._prev_t(t)
    a_t = s.alphas_cumprod[t]
    a_p = s.alphas_cumprod[pt] if pt >= 0 else s.one
    b_t, b_p = 1 - a_t, 1 - a_p
    cur_alpha = a_t / a_p
    cur_beta = 1 - cur_alpha
    x0 = s._x0(eps, t, sample)
    prev = (a_p ** 0.5 * cur_beta) / b_t * x0 + cur_alpha ** 0.5 * b_p / b_t * sample
    if t > 0:
        prev = prev + s.variance(t) ** 0.5 * z
    return osch.StepOut(prev, x0)


def _conditioning(repel: dict, x0_hat, use_beta):
    r = dict(repel)
    flavour, refs = r.pop("flavour"), r.pop("proj_refs")
    method = r.pop("method", "kernel_fast")
    if method == "sparse":
        return orp.sparse_conditioning(x0_hat, refs, flavour=flavour, radius=r["radius"], scale=r["scale"])
    return orp.kernel_fast_conditioning(x0_hat, refs, flavour=flavour, use_beta_threshold=use_beta, **r)
