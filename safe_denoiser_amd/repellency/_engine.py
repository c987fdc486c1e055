"""Shared machinery of the three repellency front-ends: proj_ref cache, workspace, libsdn launches.

The reference keeps three near-copies (repellency/repellency_methods_{threshold,fast,fast_sdv3}.py); here one
engine class carries the device state and each front-end module only fixes the semantics that differ
(which sigma is honoured, query normalisation, dtype cast, return convention, is_negation).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from .. import _lib

RBF, SPARSE = 0, 1
QNORM_NONE, QNORM_CHANNEL = 0, 1


def make_registry():
    """register_conditioning_method / get_repellency_method pair with the reference's error behaviour
    (NameError on duplicate / unknown names; repellency_methods_threshold.py:9-22)."""
    table = {}

    def register_conditioning_method(name: str):
        def wrapper(cls):
            if table.get(name, None):
                raise NameError(f"Name {name} is already registered!")
            table[name] = cls
            return cls
        return wrapper

    def get_repellency_method(name: str, ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max,
                              **kwargs):
        if table.get(name, None) is None:
            raise NameError(f"Name {name} is not defined!")
        return table[name](ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, **kwargs)

    return table, register_conditioning_method, get_repellency_method


class RepellencyEngine:
    """Device-side state shared by every method: proj_refs [M,C,H,W] fp32 resident in HBM + a workspace.

    Constructor signature = RepellencyMethod.__init__ (repellency_methods_threshold.py:25-52)."""

    float_refs = False          # fast*: project() casts to float32 (repellency_methods_fast.py:58-59)

    def __init__(self, ref_data, embed_fn, forward_fn, num_timesteps, max_idx, beta_min, beta_max, n_embed, **kwargs):
        self.ref_data, self.embed_fn, self.forward_fn = ref_data, embed_fn, forward_fn
        self.num_timesteps, self.max_idx = num_timesteps, max_idx
        self.beta_min, self.beta_max, self.n_embed = beta_min, beta_max, n_embed
        self.scale = kwargs.get("scale", 1.0)
        self.epsilon = kwargs.get("epsilon", 1e-8)
        self.proj_ref_path = kwargs.get("proj_ref_path", None)
        self.cache_proj_ref = kwargs.get("cache_proj_ref", False)
        if self.cache_proj_ref:
            self.proj_refs = self.import_proj_ref(self.proj_ref_path)
        else:
            self.proj_refs = self.set_proj_ref()
        self._ws = None
        self._ws_key = None

    # ---- proj_ref cache (row R6) ----------------------------------------------------------------
    @torch.no_grad()
    def project(self, data, **kwargs):
        """Embed in chunks of n_embed (only when len > n_embed), channel-normalise per pixel."""
        n = len(data)
        if n > self.n_embed:
            emb = torch.cat([self.embed_fn(data[i:min(i + self.n_embed, n)]) for i in range(0, n, self.n_embed)], 0)
        else:
            emb = self.embed_fn(data)
        emb = emb / torch.norm(emb, dim=1, keepdim=True)
        return emb.float() if self.float_refs else emb

    def set_proj_ref(self):
        with torch.no_grad():
            emb = self.project(self.ref_data).cpu()
        os.makedirs(os.path.split(self.proj_ref_path)[0] or ".", exist_ok=True)
        torch.save(emb, self.proj_ref_path)                 # torch.load-compatible [M,C,H,W] cache
        _lib.require_gpu()
        return emb.to("cuda")

    def import_proj_ref(self, proj_ref_path):
        return torch.load(proj_ref_path, map_location=self.ref_data.device)

    def get_proj_ref(self):
        return self.proj_refs

    def load_proj_refs(self, refs: torch.Tensor):
        """Replace the resident reference set (e.g. after an RCCL broadcast from rank 0)."""
        self.proj_refs = refs
        self._ws = None

    # ---- libsdn launches ---------------------------------------------------------------------------
    def _refs_f32(self) -> torch.Tensor:
        r = self.proj_refs
        if not r.is_cuda:
            raise _lib.SdnUnavailable("proj_refs are not on the GPU; repellency has no CPU path")
        if r.dtype != torch.float32 or not r.is_contiguous():
            r = r.float().contiguous()
            self.proj_refs = r
        return r

    def _workspace(self, n: int, refs: torch.Tensor):
        m, c, h, w = refs.shape
        key = (n, m, c, h * w, refs.device)
        if self._ws_key != key:
            nbytes = _lib.lib().sdn_repel_workspace_bytes(n, m, c, h * w)
            self._ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=refs.device)
            self._ws_key = key
        return self._ws

    def _params(self, n, refs, *, weight_fn, qnorm, sigma=1.0, radius=0.0, scale=None, gate=0.0):
        m, c, h, w = refs.shape
        return _lib.RepelParams(n_query=n, n_ref=m, channels=c, hw=h * w, weight_fn=weight_fn, qnorm=qnorm,
                                sigma=float(sigma), radius=float(radius),
                                scale=float(self.scale if scale is None else scale), epsilon=float(self.epsilon),
                                gate=float(gate))

    def apply_device(self, x: torch.Tensor, *, weight_fn=RBF, qnorm=QNORM_NONE, sigma=1.0, radius=0.0, scale=None,
                     gate=0.0, want_neg=True):
        """In-place projection of x [N,C,H,W] fp32 on the GPU.  No host synchronisation.

        Returns (neg or None, den [N] fp32, isneg [N] int32) as device tensors."""
        _lib.require_gpu()
        refs = self._refs_f32()
        if x.dtype != torch.float32 or not x.is_contiguous() or not x.is_cuda:
            raise _lib.SdnError("apply_device needs a contiguous fp32 GPU tensor (it is updated in place)")
        if tuple(x.shape[1:]) != tuple(refs.shape[1:]):
            raise _lib.SdnError(f"query shape {tuple(x.shape)} does not match proj_ref {tuple(refs.shape)}")
        n = x.shape[0]
        ws = self._workspace(n, refs)
        neg = torch.empty_like(x) if want_neg else None
        den = torch.empty(n, dtype=torch.float32, device=x.device)
        isneg = torch.empty(n, dtype=torch.int32, device=x.device)
        p = self._params(n, refs, weight_fn=weight_fn, qnorm=qnorm, sigma=sigma, radius=radius, scale=scale, gate=gate)
        _lib.check(_lib.lib().sdn_repel_apply(C.byref(p), _lib.dptr(x), _lib.dptr(refs), _lib.dptr(neg),
                                              _lib.dptr(den), _lib.dptr(isneg), _lib.dptr(ws), ws.numel(),
                                              _lib.stream_ptr()), "sdn_repel_apply")
        return neg, den, isneg

    def calibrate_device(self, queries: torch.Tensor, *, weight_fn=RBF, sigma=1.0):
        """beta[n] (RBF) or distances [N,M] (SPARSE) for calibration queries [N,C,H,W]."""
        _lib.require_gpu()
        refs = self._refs_f32()
        q = queries.float().contiguous()
        n, m = q.shape[0], refs.shape[0]
        ws = self._workspace(n, refs)
        out = torch.empty(n if weight_fn == RBF else (n, m), dtype=torch.float32, device=q.device)
        p = self._params(n, refs, weight_fn=weight_fn, qnorm=QNORM_NONE, sigma=sigma)
        _lib.check(_lib.lib().sdn_repel_calibrate(C.byref(p), _lib.dptr(q), _lib.dptr(refs), _lib.dptr(out),
                                                  _lib.dptr(ws), ws.numel(), _lib.stream_ptr()), "sdn_repel_calibrate")
        return out

    # ---- init-time calibration (row R5) ---------------------------------------------------------------
    def set_noisy_proj_ref(self, scheduler, num_timesteps=None, **kwargs):
        """For each inference timestep: add_noise(refs, randn(seed 42), t) in chunks of n_embed
        (repellency_methods_threshold.py:108-155).  Returns {t: [M,C,H,W]} on the device."""
        n_steps = num_timesteps if num_timesteps is not None else 50
        device = kwargs.get("device", "cuda")
        generator = kwargs.get("generator", None) or torch.Generator(device=device).manual_seed(42)
        refs = self._refs_f32()
        scheduler.set_timesteps(n_steps, device=device)
        out = {}
        with torch.no_grad():
            for t in scheduler.timesteps:
                parts = []
                for lo in range(0, len(refs), self.n_embed):
                    chunk = refs[lo:lo + self.n_embed].contiguous()
                    noise = torch.randn(chunk.shape, generator=generator, device=device, dtype=torch.float32)
                    parts.append(scheduler.add_noise(chunk, noise, t))
                out[t.item()] = torch.cat(parts, 0)
        path = getattr(self, "proj_beta_ref_path", None)
        if path:
            os.makedirs(os.path.split(path)[0] or ".", exist_ok=True)
            torch.save(out, path)
        return out
