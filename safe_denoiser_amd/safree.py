"""SAFREE text-side projection: the once-per-prompt preprocessing that produces the
`rescaled_text_embeddings` and the self-validation-filter step count the denoising loop consumes.

Restates models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:19-99 (f_beta,
projection_matrix, projection_and_orthogonal, safree_projection) and :458-486 (call order, beta).  It runs ONCE per
prompt on [<=77, 768] matrices (the reference runs it with torch on the GPU too); it is host orchestration outside the
step loop, so it is written with torch ops here -- the hot path proper stays in libsdn.  The CLIP encodes it needs
(`masked_embs` = pooled embedding of the prompt with each token masked in turn, `negspace` = pooled embeddings of the
negative-concept phrases) are produced by SafeDenoiserPipeline._masked_encode_prompt / _new_encode_negative_prompt_space
with the engine's CLIPTextModel (pipeline.py), or passed in by a caller that has its own text encoder.
Pinned: tests/golden/safree_golden.npz holds outputs of the reference's own helpers (tests/test_safree.py).
"""
from __future__ import annotations

import math

import torch


def f_beta(z: float, btype: str = "sigmoid", upperbound_timestep: int = 10, concept_type: str = "nudity") -> int:
    """Self-validation filter: number of leading steps that use the projected embeddings (:19-36)."""
    t, k = (5.5, 3.5) if "artists-" in concept_type else (5.333, 2.5)
    if btype == "tanh":
        return round(upperbound_timestep / 2.0 * (math.tanh(k * (10 * z - t)) + 1))
    if btype == "sigmoid":
        return round(upperbound_timestep * (1.0 / (1.0 + math.exp(-2.0 * k * (10 * z - t)))))
    raise NotImplementedError("btype is incorrect")


def projection_matrix(E: torch.Tensor) -> torch.Tensor:
    """Projector onto span(columns of E): E (E^T E)^+ E^T (:38-41)."""
    return E @ torch.pinverse(E.T @ E) @ E.T


def safree_projection(input_embeddings: torch.Tensor, p_emb: torch.Tensor, masked_proj: torch.Tensor,
                      concept_proj: torch.Tensor, alpha: float = 0.0, max_length: int = 77):
    """Token-wise replacement of trigger tokens by (I - P_c) P_m e (:56-99).
    input_embeddings [2,77,768] (uncond, text); p_emb [n_t,768]; returns ([2,77,768], n_removed)."""
    n_t, dim = p_emb.shape
    eye_m_c = torch.eye(dim, device=p_emb.device, dtype=p_emb.dtype) - concept_proj
    dist = torch.norm(eye_m_c @ p_emb.T, dim=0)                              # distance of each masked prompt to the concept space
    loo_mean = (dist.sum() - dist) / (n_t - 1) if n_t > 1 else torch.full_like(dist, float("nan"))
    keep = (dist < (1.0 + alpha) * loo_mean).float()                         # 1 = safe token, 0 = trigger token
    mask = torch.ones(max_length, device=p_emb.device, dtype=p_emb.dtype)
    mask[1:n_t + 1] = keep
    uncond_e, text_e = input_embeddings.chunk(2)
    text_e = text_e.squeeze(0)
    projected = (eye_m_c @ masked_proj @ text_e.T).T
    merged = torch.where(mask.bool()[:, None], text_e, projected)
    return torch.cat([uncond_e, merged.unsqueeze(0)]), int(n_t - keep.sum())


def token_keep_mask(text_e: torch.Tensor, rescaled_e: torch.Tensor) -> torch.Tensor:
    """[.., L] bool: True where the projection left the token's embedding as it was (the `mask` of safree_projection)."""
    return (text_e == rescaled_e).all(dim=-1)


def projection_and_orthogonal(input_embeddings, masked_proj, concept_proj):
    """(I - P_c) P_m applied to every token (:44-54)."""
    dim = masked_proj.shape[0]
    uncond_e, text_e = input_embeddings.chunk(2)
    new_text = ((torch.eye(dim, device=text_e.device, dtype=text_e.dtype) - concept_proj) @ masked_proj
                @ text_e.squeeze(0).T).T[None]
    return torch.cat([uncond_e, new_text])


def prepare(text_embeddings: torch.Tensor, masked_embs: torch.Tensor, negspace: torch.Tensor,
            attention_mask: torch.Tensor, *, alpha: float = 0.01, svf: bool = True, up_t: int = 10,
            category: str = "nudity", concept_proj: torch.Tensor | None = None) -> dict:
    """The reference's call sequence (:458-486).  Returns what SafeDenoiserPipeline takes:
    rescaled_text_embeddings, beta_adjusted (None when svf is off), plus diagnostics.  `concept_proj`: the projector of
    `negspace` when the caller already has it (it is the same for every prompt of a batch)."""
    P_c = projection_matrix(negspace.T) if concept_proj is None else concept_proj
    P_m = projection_matrix(masked_embs.T)
    rescaled, n_removed = safree_projection(text_embeddings, masked_embs, P_m, P_c, alpha=alpha,
                                            max_length=text_embeddings.shape[1])
    out = {"rescaled_text_embeddings": rescaled, "n_removed": n_removed, "beta_adjusted": None, "beta": None,
           "token_mask": token_keep_mask(text_embeddings[1], rescaled[1])}
    if svf:
        proj_ort = projection_and_orthogonal(text_embeddings, P_m, P_c)
        act = attention_mask.reshape(-1) == 1
        text_e = text_embeddings[1][act]
        po = proj_ort[1][act]
        beta = 1.0 - float(torch.nn.functional.cosine_similarity(po, text_e).mean())
        out["beta"] = beta
        out["beta_adjusted"] = f_beta(beta, upperbound_timestep=up_t, concept_type=category)
    return out


def prepare_batch(text_embeddings: torch.Tensor, masked_embs: list, negspace: torch.Tensor, attention_masks: torch.Tensor, *,
                  alpha: float = 0.01, svf: bool = True, up_t: int = 10, category: str = "nudity",
                  concept_proj: torch.Tensor | None = None) -> dict:
    """`prepare` for P prompts at once (the batched engine's front end; run_nudity.py calls the pipeline one prompt at a time).
    text_embeddings [2P,77,768] ([uncond | text] rows), masked_embs: P tensors [n_t(p), 768], attention_masks [P,77].
    Same arithmetic per prompt as `prepare` -- P_m = E (E^T E)^+ E^T from the SVD pseudo-inverse, the leave-one-out distance
    test, the token-wise replacement, the cosine statistic -- with the prompts' masked-embedding matrices zero-padded to a common
    token count (zero rows change neither the Gram matrix's range nor its pseudo-inverse on it), so that the 64 small SVDs,
    projector products and reductions of a batch are one batched call each instead of ~15 launches and two host syncs per
    prompt.  The masked-prompt Gram matrices are near-singular (the rows are one prompt with one token changed), and the
    reference's rcond of 1e-15 inverts whatever singular values rounding left there; a BATCHED fp32 SVD on padded matrices
    rounds differently from the per-prompt one, which can move a trigger-token test or f_beta's round across its edge
    (ADVICE r3).  So the Gram matrix, its pseudo-inverse and the projector P_m are formed in float64 here (P x n x n, n <= 75:
    microseconds) -- the exact projector the reference's fp32 chain approximates -- and cast to the working dtype once.
    Returns the per-prompt lists `prepare` would give, stacked."""
    P = len(masked_embs)
    dev, dt = text_embeddings.device, text_embeddings.dtype
    dim, L = text_embeddings.shape[-1], text_embeddings.shape[1]
    P_c = projection_matrix(negspace.T) if concept_proj is None else concept_proj
    eye_m_c = torch.eye(dim, device=dev, dtype=dt) - P_c
    counts = [int(m.shape[0]) for m in masked_embs]
    n_max = max(max(counts), 1)
    Mp = torch.zeros((P, n_max, dim), device=dev, dtype=dt)
    valid = torch.zeros((P, n_max), device=dev, dtype=torch.bool)
    for p_, m in enumerate(masked_embs):
        Mp[p_, :counts[p_]] = m
        valid[p_, :counts[p_]] = True
    Md = Mp.double()
    G = Md @ Md.transpose(1, 2)                                              # [P, n, n] = E^T E of each prompt (zero-padded), float64
    P_m = (Md.transpose(1, 2) @ torch.linalg.pinv(G, rtol=1e-15, hermitian=True) @ Md).to(dt)   # [P, dim, dim]; torch.pinverse's cut-off (rcond 1e-15)
    text_e = text_embeddings[P:]                                             # [P, L, dim]
    # distance of each masked prompt to the concept space, leave-one-out mean test (safree_projection)
    dist = torch.linalg.vector_norm(Mp @ eye_m_c.T, dim=-1)                  # [P, n]   (= ||(I - P_c) p_emb^T|| per column)
    dist = torch.where(valid, dist, torch.zeros_like(dist))
    n_t = valid.sum(dim=1, keepdim=True).to(dt)
    loo = (dist.sum(dim=1, keepdim=True) - dist) / (n_t - 1)
    loo = torch.where(n_t > 1, loo, torch.full_like(loo, float("nan")))
    keep = (dist < (1.0 + alpha) * loo) & valid                              # True = safe token
    mask = torch.ones((P, L), device=dev, dtype=torch.bool)
    mask[:, 1:n_max + 1] = keep | ~valid                                     # positions past a prompt's tokens keep their embedding
    projected = (text_e @ P_m.transpose(1, 2)) @ eye_m_c.T                   # ((I - P_c) P_m e)^T for every token
    merged = torch.where(mask[:, :, None], text_e, projected)
    n_removed = (valid & ~keep).sum(dim=1)
    out = {"rescaled_text_embeddings": torch.cat([text_embeddings[:P], merged]), "n_removed": n_removed.tolist(),
           "beta": [None] * P, "beta_adjusted": [None] * P, "token_mask": mask}
    if svf:
        act = (attention_masks.to(dev) == 1).to(dt)                          # [P, L]
        cos = torch.nn.functional.cosine_similarity(projected, text_e, dim=-1)
        beta = 1.0 - (cos * act).sum(dim=1) / act.sum(dim=1)
        out["beta"] = [float(b) for b in beta.tolist()]
        out["beta_adjusted"] = [f_beta(b, upperbound_timestep=up_t, concept_type=category) for b in out["beta"]]
    return out


# ---- SD-v3 variant (models/sdv3/safe_denoiser_pipeline.py:72-153,1061-1078): T5 hidden states, 16-bit matmuls ------------
def projection_matrix_sd3(E: torch.Tensor) -> torch.Tensor:
    """:72-82: the same projector, computed in fp32 whatever the embeddings' dtype."""
    Ef = E.float()
    return Ef @ torch.pinverse(Ef.T @ Ef) @ Ef.T


def mask_to_onp(input_embeddings: torch.Tensor, p_emb: torch.Tensor, masked_proj: torch.Tensor, concept_proj: torch.Tensor,
                alpha: float = 0.0, max_length: int = 333):
    """:100-153.  As safree_projection, with the reference's dtype quirks kept: the two matrix products run in bfloat16
    ((I - P_c).bfloat16() @ p_emb.T.bfloat16(), and (I - P_c).bf16 @ P_m.bf16 @ text.T.bf16), the distance norm and the
    leave-one-out means in that dtype's values; the token mask is written at positions 1..n_t of a length-`max_length` vector
    (the 333 = 77 CLIP + 256 T5 token axis).  Returns (embeddings [2, max_length, dim], keep mask [max_length, 1],
    inverse mask [max_length], n_removed)."""
    n_t, dim = p_emb.shape
    dev = p_emb.device
    eye_m_c = torch.eye(dim, device=dev) - concept_proj
    dist = torch.norm(eye_m_c.bfloat16() @ p_emb.T.bfloat16(), dim=0)
    means = [torch.mean(torch.cat((dist[:i], dist[i + 1:]))) for i in range(n_t)]
    mean_dist = torch.tensor(means).to(dev)                                   # float32 copies of the bf16 means, as torch.tensor() makes
    keep = (dist < (1.0 + alpha) * mean_dist).float()
    inv = (dist >= (1.0 + alpha) * mean_dist).float()
    n_removed = n_t - keep.sum()
    ones = torch.ones(max_length, device=dev)
    ones[1:n_t + 1] = keep
    ones = ones.unsqueeze(1)
    inverse = torch.ones(max_length, device=dev)
    inverse[1:n_t + 1] = inv
    uncond_e, text_e = input_embeddings.chunk(2)
    text_e = text_e.squeeze()
    projected = (eye_m_c.bfloat16() @ masked_proj.bfloat16() @ text_e.T.bfloat16()).T
    merged = torch.where(ones.bool(), text_e, projected)
    return torch.cat([uncond_e, merged.unsqueeze(0)]), ones, inverse, n_removed.item()


def prepare_sd3(text_embeddings: torch.Tensor, masked_embs: torch.Tensor, negspace: torch.Tensor, *, alpha: float = 0.01) -> dict:
    """The call sequence of the SD-v3 pipeline (:1061-1078): masked_embs = first-token T5 states of the prompt with each
    token masked in turn, negspace = first-token T5 states of the concept phrases; the loop then feeds the result to the
    transformer at EVERY step (:1115)."""
    P_m = projection_matrix_sd3(masked_embs.T)
    P_c = projection_matrix_sd3(negspace.T)
    resc, keep, inv, n_removed = mask_to_onp(text_embeddings, masked_embs, P_m, P_c, alpha=alpha, max_length=text_embeddings.shape[1])
    return {"rescaled_text_embeddings": resc, "sp_vector": keep, "inv_vector": inv, "n_removed": n_removed}
