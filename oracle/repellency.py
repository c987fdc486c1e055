"""CPU oracle: repellency projection.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Pinned against golden vectors
captured from the reference's own modules (tests/golden/repellency_golden.npz).

The reference keeps three modules that register a class under the same key
``kernel_fast`` with different semantics.  They are restated here as plain
functions selected by a ``flavour`` string:

  "threshold"  -> repellency/repellency_methods_threshold.py
  "fast"       -> repellency/repellency_methods_fast.py
  "fast_sdv3"  -> repellency/repellency_methods_fast_sdv3.py

All functions take and return torch CPU tensors.  Where the reference mutates
its argument in place the oracle does too (and says so), because callers in the
reference rely on that aliasing (SURVEY.md section 3.2 "Interaction trap").
"""
from __future__ import annotations

import torch

FLAVOURS = ("threshold", "fast", "fast_sdv3")


# --------------------------------------------------------------------------
# R6: building proj_ref
# --------------------------------------------------------------------------
def channel_normalise(z: torch.Tensor) -> torch.Tensor:
    """z / ||z||_2 over dim=1 per pixel.

    repellency_methods_threshold.py:63-64,70-71; fast:55-56,66-67;
    fast_sdv3 applies the same op to the *query* (fast_sdv3:239).
    """
    return z / torch.linalg.vector_norm(z, ord=2, dim=1, keepdim=True)


def project_refs(data: torch.Tensor, embed_fn, n_embed: int, flavour: str = "threshold") -> torch.Tensor:
    """Embed negatives in chunks of n_embed, then channel-normalise.

    threshold:54-72 keeps the embed dtype; fast:45-72 additionally casts to
    float32.  Chunking only happens when len(data) > n_embed (``>``, not ``>=``).
    """
    n = len(data)
    if n > n_embed:
        pieces = [embed_fn(data[lo:min(lo + n_embed, n)]) for lo in range(0, n, n_embed)]
        emb = torch.cat(pieces, 0)
    else:
        emb = embed_fn(data)
    emb = channel_normalise(emb)
    if flavour != "threshold":
        emb = emb.float()
    return emb


# --------------------------------------------------------------------------
# R1/R2/R3: kernel_fast empirical denoiser
# --------------------------------------------------------------------------
def rbf_weights(xq: torch.Tensor, refs: torch.Tensor, sigma: float) -> torch.Tensor:
    """w[n,m] = exp(-||x_n - r_m||_2 / (2 sigma^2)) -- UN-squared distance.

    threshold:335 / fast:249 (``torch.cdist`` default p=2, not squared).
    xq [N,D], refs [M,D] -> [N,M].
    """
    dist = torch.cdist(xq[None], refs[None])[0]
    return torch.exp(-dist / (2.0 * sigma ** 2))


def kernel_fast_score(x: torch.Tensor, proj_refs: torch.Tensor, sigma: float, epsilon: float,
                      query_channel_norm: bool = False):
    """Kernel-weighted mean of reference latents.

    threshold:309-349, fast:223-262, fast_sdv3:229-271.
    Returns (neg [N,C,H,W], denominator [N], numerator [N,D]).
    No max-subtraction; additive epsilon in the denominator (reproduced, not
    "fixed": with sigma=1 the weights underflow and epsilon dominates).
    """
    if query_channel_norm:                     # fast_sdv3:238-240 (local rebinding only)
        x = channel_normalise(x)
    n = x.shape[0]
    m, c, h, w_ = proj_refs.shape
    xq = x.reshape(n, -1)
    refs = proj_refs.reshape(m, -1)
    wts = rbf_weights(xq, refs, sigma)                                  # [N,M]
    aug = torch.cat((refs, torch.ones(m, 1, dtype=refs.dtype, device=refs.device)), dim=1)  # [M,D+1]
    acc = (wts[:, :, None] * aug[None]).sum(dim=1)                      # [N,D+1]
    den = acc[:, -1] + epsilon
    num = acc[:, :-1]
    neg = (num / den[:, None]).reshape(n, c, h, w_)
    return neg, den, num


def kernel_fast_conditioning(x_0_hat: torch.Tensor, proj_refs: torch.Tensor, *, flavour: str,
                             scale: float, epsilon: float = 1e-8, sigma: float = 1.0,
                             beta_threshold: float = -1.0, beta_threshold_margin: float = 0.0,
                             use_beta_threshold: bool = False, guidance_scale=None,
                             ref_dtype: torch.dtype = torch.float32) -> dict:
    """``RBFKernelRepellency.conditioning`` for all three modules.

    threshold (conditioning :171-175):
      use_beta_threshold=True  -> conditioning_threshold :177-188 -- x -= scale*neg
          IN PLACE, returns x itself, is_negation = den > beta_threshold - margin.
      use_beta_threshold=False -> conditioning_1 :190-193 -- same in-place update
          but RETURNS THE NEGATIVE SCORE as "x_0_hat"; is_negation always True.
      sigma is the YAML value (threshold:36,179,191).
    fast / fast_sdv3 (conditioning :120-127):
      x is first cast to the refs' dtype (a copy if the dtype differs -> the
      caller's tensor is then NOT mutated); sigma is ALWAYS 1.0 (the YAML value
      is never read: fast:24-43,129-132,223);
      guidance_scale > 0 -> conditioning_2 :134-137: x -= neg (no scale), returns neg;
      else conditioning_1 :129-132: x -= scale*neg, returns x.  No is_negation key.
    Only N == 1 is legal in the reference (``denominator.item()``, threshold:348).
    """
    assert flavour in FLAVOURS
    if flavour == "threshold":
        neg, den, num = kernel_fast_score(x_0_hat, proj_refs, sigma, epsilon)
        item = {"negative_score_item": float(neg.clamp(min=-1e10, max=1e10).mean()),
                "denominator": float(den.reshape(-1)[0]) if den.numel() == 1 else den.clone(),
                "nominator": num}
        x_0_hat -= scale * neg
        if use_beta_threshold:
            gate = beta_threshold - beta_threshold_margin
            is_neg = bool(den.reshape(-1)[0] > gate) if den.numel() == 1 else (den > gate)
            return {"x_0_hat": x_0_hat, "mean_x_0_hat": item, "is_negation": is_neg}
        return {"x_0_hat": neg, "mean_x_0_hat": item, "is_negation": True}

    # fast / fast_sdv3
    if x_0_hat.dtype != ref_dtype:
        x_0_hat = x_0_hat.to(ref_dtype)
    neg, _den, _num = kernel_fast_score(x_0_hat, proj_refs, 1.0, epsilon,
                                        query_channel_norm=(flavour == "fast_sdv3"))
    item = float(neg.clamp(min=-1e10, max=1e10).mean())
    if guidance_scale is not None and guidance_scale > 0.0:
        x_0_hat -= neg
        return {"x_0_hat": neg, "mean_x_0_hat": item}
    x_0_hat -= scale * neg
    return {"x_0_hat": x_0_hat, "mean_x_0_hat": item}


# --------------------------------------------------------------------------
# R4: sparse (SPELL-style) repellency
# --------------------------------------------------------------------------
def sparse_force(x: torch.Tensor, proj_refs: torch.Tensor, radius: float, query_channel_norm: bool = False):
    """Sum over refs within ``radius`` of (x - r) * relu(radius/||x-r|| - 1).

    threshold:415-439, fast:306-329 (fast_sdv3:331-333 normalises the query).
    x [1,C,H,W].  Returns (force [1,C,H,W], trunc_weight [1,K] over the K kept refs).
    """
    if query_channel_norm:
        x = channel_normalise(x)
    dist = torch.linalg.vector_norm(x - proj_refs, dim=(1, 2, 3))
    keep = dist < radius
    near = proj_refs[keep]
    diff = x.unsqueeze(1) - near.unsqueeze(0)                     # [1,K,C,H,W]
    wnorm = torch.linalg.vector_norm(diff, dim=(2, 3, 4))           # empty-neighbour case -> [1,0]
    trunc = torch.relu(radius / wnorm - 1.0)                      # [1,K]
    force = (diff * trunc[..., None, None, None]).sum(dim=1)
    return force, trunc


def sparse_conditioning(x_0_hat: torch.Tensor, proj_refs: torch.Tensor, *, flavour: str,
                        radius: float, scale: float, ref_dtype: torch.dtype = torch.float32) -> dict:
    """threshold:446-459 (is_negation = any weight > 0) / fast:336-340 (no key).

    In-place ``x += scale * force``; returns x itself.
    """
    if flavour != "threshold" and x_0_hat.dtype != ref_dtype:
        x_0_hat = x_0_hat.to(ref_dtype)
    force, trunc = sparse_force(x_0_hat, proj_refs, radius, query_channel_norm=(flavour == "fast_sdv3"))
    x_0_hat += scale * force
    out = {"x_0_hat": x_0_hat, "mean_x_0_hat": float(torch.linalg.vector_norm(force))}
    if flavour == "threshold":
        out["is_negation"] = bool(trunc.sum() != 0.0)
    return out


# --------------------------------------------------------------------------
# R5: init-time calibration
# --------------------------------------------------------------------------
def make_noisy_refs(proj_refs: torch.Tensor, add_noise, timesteps, n_embed: int,
                    generator: torch.Generator) -> dict:
    """threshold:108-155: for each t, add_noise(refs, randn, t) in chunks of n_embed.

    One generator stream (seed 42 in the reference) is consumed in (t, chunk) order.
    """
    out = {}
    for t in timesteps:
        parts = []
        for lo in range(0, len(proj_refs), n_embed):
            chunk = proj_refs[lo:lo + n_embed]
            noise = torch.randn(chunk.shape, generator=generator, dtype=torch.float32)
            parts.append(add_noise(chunk, noise, t))
        out[int(t)] = torch.cat(parts, 0)
    return out


def empirical_beta(noisy_refs: dict, proj_refs: torch.Tensor, sigma: float, epsilon: float, q: float) -> dict:
    """threshold:351-384: beta_n = sum_m exp(-||x_n - r_m||/(2 sigma^2)) + eps; quantile over n."""
    refs = proj_refs.reshape(proj_refs.shape[0], -1)
    out = {}
    for t, lat in noisy_refs.items():
        beta = rbf_weights(lat.reshape(lat.shape[0], -1), refs, sigma).sum(dim=1) + epsilon
        out[t] = torch.quantile(beta, q)
    return out


def empirical_radius(noisy_refs: dict, proj_refs: torch.Tensor, q: float) -> dict:
    """threshold:461-490: quantile of all N*M pairwise L2 distances (direct differences)."""
    refs = proj_refs.reshape(proj_refs.shape[0], -1)
    out = {}
    for t, lat in noisy_refs.items():
        rows = [torch.linalg.vector_norm(row[None] - refs, dim=1) for row in lat.reshape(lat.shape[0], -1)]
        out[t] = torch.quantile(torch.cat(rows, 0), q)
    return out


def calibrated_threshold(per_t: dict):
    """The reference keeps the LAST key of the dict (t -> 1), threshold:302,409."""
    return per_t[list(per_t.keys())[-1]]
