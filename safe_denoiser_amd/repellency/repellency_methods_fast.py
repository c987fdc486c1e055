"""Front-end with the semantics of the reference's repellency/repellency_methods_fast.py (run_copro.py:52).

Differences from the threshold module that are reproduced on purpose:
  * YAML `sigma` is never read -> the kernel bandwidth is ALWAYS 1.0 (fast:24-43,129-132,223);
  * x is first cast to the references' dtype (a COPY when the dtype differs -> the caller's tensor is then
    not mutated and the returned tensor is the fp32 copy) (fast:120-122);
  * no "is_negation" key; `guidance_scale > 0` selects conditioning_2: x -= neg (no scale), returns neg.
"""
from __future__ import annotations

import torch

from ._engine import QNORM_CHANNEL, QNORM_NONE, RBF, SPARSE, RepellencyEngine, make_registry

__CONDITIONING_METHOD__, register_conditioning_method, get_repellency_method = make_registry()


class RepellencyMethod(RepellencyEngine):
    float_refs = True
    qnorm = QNORM_NONE            # fast_sdv3 overrides

    def _cast(self, x_0_hat):
        if x_0_hat.dtype != self.ref_data.dtype:
            x_0_hat = x_0_hat.to(self.ref_data.dtype)
        if x_0_hat.dtype != torch.float32:
            raise NotImplementedError("the resident proj_ref is fp32; ref_data.dtype must be float32")
        return x_0_hat.contiguous()

    def conditioning(self, x_0_hat, **kwargs):
        x_0_hat = self._cast(x_0_hat)
        g = kwargs.get("guidance_scale", None)
        if g is not None and g > 0.0:
            return self.conditioning_2(x_0_hat, **kwargs)
        return self.conditioning_1(x_0_hat, **kwargs)


@register_conditioning_method(name="kernel_fast")
class RBFKernelRepellency(RepellencyMethod):
    """repellency_methods_fast.py:217-262."""

    def __init__(self, ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, **kwargs):
        super().__init__(ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, **kwargs)
        self.scale = kwargs.get("scale", 1.0)

    def conditioning_device(self, x_0_hat, want_neg=False, scale=None, **unused):
        return self.apply_device(x_0_hat, weight_fn=RBF, qnorm=self.qnorm, sigma=1.0, scale=scale,
                                 gate=float("inf"), want_neg=want_neg)

    def empirical_denoiser(self, x_t, sigma=1.0, **kwargs):
        x = x_t.clone()
        neg, _den, _ = self.apply_device(x, weight_fn=RBF, qnorm=self.qnorm, sigma=sigma, scale=0.0)
        return neg, neg.clamp(min=-1e10, max=1e10).mean().item()

    def conditioning_1(self, x_0_hat, **kwargs):
        neg, _den, _ = self.conditioning_device(x_0_hat, want_neg=True)
        return {"x_0_hat": x_0_hat, "mean_x_0_hat": neg.clamp(min=-1e10, max=1e10).mean().item()}

    def conditioning_2(self, x_0_hat, **kwargs):
        neg, _den, _ = self.conditioning_device(x_0_hat, want_neg=True, scale=1.0)
        return {"x_0_hat": neg, "mean_x_0_hat": neg.clamp(min=-1e10, max=1e10).mean().item()}


@register_conditioning_method(name="sparse")
class SparseRepellency(RepellencyMethod):
    """repellency_methods_fast.py:299-340."""

    def __init__(self, ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, **kwargs):
        super().__init__(ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, **kwargs)
        self.radius = kwargs.get("radius", 1.0)
        self.scale = kwargs.get("scale", 1.0)

    def conditioning_device(self, x_0_hat, want_neg=False, **unused):
        return self.apply_device(x_0_hat, weight_fn=SPARSE, qnorm=self.qnorm, radius=float(self.radius),
                                 want_neg=want_neg)

    def repellency_force(self, x_0_hat, **kwargs):
        x = x_0_hat.clone()
        force, _s, _ = self.apply_device(x, weight_fn=SPARSE, qnorm=self.qnorm, radius=float(self.radius), scale=0.0)
        return force, force.norm(p=2).item()

    empirical_denoiser = repellency_force

    def conditioning_1(self, x_0_hat, **kwargs):
        force, _s, _ = self.conditioning_device(x_0_hat, want_neg=True)
        return {"x_0_hat": x_0_hat, "mean_x_0_hat": force.norm(p=2).item()}

    conditioning_2 = conditioning_1


@register_conditioning_method(name="random_noise")
class RandomNoiseRepellency(RepellencyMethod):
    """repellency_methods_fast.py:264-297: the 'negative score' is a fresh randn (global RNG); shape/return
    convention only -- there is no arithmetic to accelerate."""

    def __init__(self, ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, **kwargs):
        super().__init__(ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, **kwargs)
        self.scale = kwargs.get("scale", 1.0)

    def empirical_denoiser(self, x_t, sigma=1.0, **kwargs):
        _, c, h, w = self.proj_refs.shape
        neg = torch.randn(size=(1, c * h * w)).to(self.proj_refs.device).reshape(-1, c, h, w)
        return neg, neg.clamp(min=-1e10, max=1e10).mean().item()

    def conditioning_1(self, x_0_hat, **kwargs):
        neg, item = self.empirical_denoiser(x_0_hat)
        x_0_hat -= self.scale * neg
        return {"x_0_hat": x_0_hat, "mean_x_0_hat": item}

    def conditioning_2(self, x_0_hat, **kwargs):
        neg, item = self.empirical_denoiser(x_0_hat)
        x_0_hat -= neg
        return {"x_0_hat": neg, "mean_x_0_hat": item}
