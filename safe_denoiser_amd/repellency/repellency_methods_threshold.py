"""Front-end with the semantics of the reference's repellency/repellency_methods_threshold.py
(used by run_nudity.py:53-54, run_coco30k.py:55, run_munch.py:48, run_ann_graham.py:46).

    get_repellency_method(name, ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, **kw)
    proc.conditioning(x_0_hat, beta_threshold=bool) -> {"x_0_hat", "mean_x_0_hat", "is_negation"}

Semantics kept: YAML sigma IS honoured here; un-squared distance; additive epsilon;
x_0_hat mutated in place; beta_threshold=False returns the NEGATIVE SCORE as "x_0_hat" with is_negation=True.
Extension: N > 1 queries per call (the reference throws on .item()); is_negation is then a bool tensor.
`conditioning_device` is the sync-free form the batched engine loop uses.
"""
from __future__ import annotations

import torch

from ._engine import QNORM_NONE, RBF, SPARSE, RepellencyEngine, make_registry

__CONDITIONING_METHOD__, register_conditioning_method, get_repellency_method = make_registry()


class RepellencyMethod(RepellencyEngine):
    def __init__(self, ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, n_embed, **kwargs):
        self.sigma = kwargs.get("sigma", 1.0)
        self.quantile = kwargs.get("quantile", 0.0)
        self.beta_threshold = kwargs.get("beta_threshold", False)
        self.beta_threshold_margin = kwargs.get("beta_threshold_margin", 0.0)
        self.proj_beta_ref_path = kwargs.get("proj_noisy_ref_path_for_beta", None)
        self.cache_proj_beta_ref = kwargs.get("cache_noisy_ref_path_for_beta", False)
        super().__init__(ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, n_embed, **kwargs)

    def get_noisy_proj_refs(self):
        return self.noisy_proj_refs

    def conditioning(self, x_0_hat, **kwargs):
        if kwargs.get("beta_threshold", False):
            return self.conditioning_threshold(x_0_hat, **kwargs)
        return self.conditioning_1(x_0_hat, **kwargs)

    # -- host-visible dict assembly (does the .item() syncs the reference does) -------------------------
    def _item(self, neg, den):
        num = neg.reshape(neg.shape[0], -1) * den[:, None]
        d = den.item() if den.numel() == 1 else den
        return {"negative_score_item": neg.clamp(min=-1e10, max=1e10).mean().item(), "denominator": d,
                "nominator": num}

    @staticmethod
    def _flag(isneg):
        return bool(isneg.item()) if isneg.numel() == 1 else isneg.bool()


@register_conditioning_method(name="kernel_fast")
class RBFKernelRepellency(RepellencyMethod):
    """repellency_methods_threshold.py:282-384."""

    def __init__(self, ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, **kwargs):
        super().__init__(ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, **kwargs)
        self.scale = kwargs.get("scale", 1.0)
        self.beta_threshold = kwargs.get("beta_threshold", -1.0)
        if self.beta_threshold <= 0:
            if self.cache_proj_beta_ref:
                self.noisy_proj_refs = self.import_proj_ref(self.proj_beta_ref_path)
            else:
                scheduler = kwargs.get("scheduler", None)
                assert scheduler is not None, "We need scheduler for computing beta reference"
                self.noisy_proj_refs = self.set_noisy_proj_ref(scheduler, self.num_timesteps)
            self.noisy_refs_beta_quantitle = self.empirical_beta(sigma=self.sigma, quantitle=self.quantile)
            # the reference keeps the LAST key (smallest t), :302
            # (stored ONCE as a host float: conditioning_device() builds the gate from it every step and must not sync)
            self.beta_threshold = float(self.noisy_refs_beta_quantitle[list(self.noisy_refs_beta_quantitle.keys())[-1]].item())
            del self.noisy_proj_refs, self.noisy_refs_beta_quantitle

    def conditioning_device(self, x_0_hat, beta_threshold=True, want_neg=False):
        """Sync-free: x updated in place; returns (neg|None, den[N], isneg[N] int32) device tensors."""
        gate = float(self.beta_threshold) - float(self.beta_threshold_margin) if beta_threshold else float("-inf")
        neg, den, isneg = self.apply_device(x_0_hat, weight_fn=RBF, qnorm=QNORM_NONE, sigma=self.sigma, gate=gate,
                                            want_neg=want_neg)
        if not beta_threshold:
            isneg.fill_(1)      # conditioning_1 reports is_negation=True unconditionally (:190-193), also when den is NaN
        return neg, den, isneg

    def empirical_denoiser(self, x_t, sigma=1.0, **kwargs):
        x = x_t.clone()
        neg, den, _ = self.apply_device(x, weight_fn=RBF, sigma=sigma, scale=0.0)
        return neg, self._item(neg, den)

    def conditioning_threshold(self, x_0_hat, **kwargs):
        neg, den, isneg = self.conditioning_device(x_0_hat, beta_threshold=True, want_neg=True)
        return {"x_0_hat": x_0_hat, "mean_x_0_hat": self._item(neg, den), "is_negation": self._flag(isneg)}

    def conditioning_1(self, x_0_hat, **kwargs):
        neg, den, _ = self.conditioning_device(x_0_hat, beta_threshold=False, want_neg=True)
        return {"x_0_hat": neg, "mean_x_0_hat": self._item(neg, den), "is_negation": True}

    def empirical_beta(self, sigma=1.0, quantitle=0.25, **kwargs):
        """{t: quantile_n( sum_m exp(-||noisy_n - r_m||/(2 sigma^2)) + eps )}  (:351-384)."""
        out = {}
        for t, latents in self.get_noisy_proj_refs().items():
            beta = self.calibrate_device(latents.to(self.proj_refs.device), weight_fn=RBF, sigma=sigma)
            out[t] = torch.quantile(beta, quantitle)
        return out


@register_conditioning_method(name="sparse")
class SparseRepellency(RepellencyMethod):
    """repellency_methods_threshold.py:386-490 (SPELL-style)."""

    def __init__(self, ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, **kwargs):
        super().__init__(ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, **kwargs)
        self.radius = kwargs.get("radius", -1.0)
        self.scale = kwargs.get("scale", 1.0)
        if self.radius <= 0:
            if self.cache_proj_beta_ref:
                self.noisy_proj_refs = self.import_proj_ref(self.proj_beta_ref_path)
            else:
                scheduler = kwargs.get("scheduler", None)
                assert scheduler is not None, "We need scheduler for computing beta reference"
                self.noisy_proj_refs = self.set_noisy_proj_ref(scheduler, self.num_timesteps)
            self.noisy_refs_beta_quantitle = self.empirical_radius(quantitle=self.quantile)
            self.radius = self.noisy_refs_beta_quantitle[list(self.noisy_refs_beta_quantitle.keys())[-1]]
            del self.noisy_proj_refs, self.noisy_refs_beta_quantitle

    def conditioning_device(self, x_0_hat, beta_threshold=True, want_neg=False):
        return self.apply_device(x_0_hat, weight_fn=SPARSE, radius=float(self.radius), want_neg=want_neg)

    def repellency_force(self, x_0_hat, **kwargs):
        x = x_0_hat.clone()
        force, sumw, _ = self.apply_device(x, weight_fn=SPARSE, radius=float(self.radius), scale=0.0)
        return force, {"repellency_force": force.norm(p=2).item(), "trunc_weight": sumw}

    empirical_denoiser = repellency_force

    def conditioning_1(self, x_0_hat, **kwargs):
        force, _sumw, isneg = self.conditioning_device(x_0_hat, want_neg=True)
        return {"x_0_hat": x_0_hat, "mean_x_0_hat": force.norm(p=2).item(), "is_negation": self._flag(isneg)}

    def conditioning_threshold(self, x_0_hat, **kwargs):
        return self.conditioning_1(x_0_hat, **kwargs)

    def empirical_radius(self, quantitle=0.25, **kwargs):
        out = {}
        for t, latents in self.get_noisy_proj_refs().items():
            dist = self.calibrate_device(latents.to(self.proj_refs.device), weight_fn=SPARSE)
            out[t] = torch.quantile(dist.reshape(-1), quantitle)
        return out
