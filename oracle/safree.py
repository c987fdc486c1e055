"""CPU oracle (numpy, independent of torch's pinverse) for the SAFREE text projection
(models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:38-99,458-486).  TEST INFRASTRUCTURE.
Parity status: PINNED -- the module that holds these helpers imports diffusers at file scope and cannot be imported, but the
helper functions themselves are pure torch: tests/golden/make_safree_golden.py parses the file, executes exactly those function
definitions and stores their inputs / outputs (tests/golden/safree_golden.npz); tests/test_safree.py holds this restatement
and the product's safree.py to them."""
import numpy as np


def proj(E):
    return E @ np.linalg.pinv(E.T @ E) @ E.T


def safree(ie, p_emb, alpha, max_length=77):
    """ie [2,77,768] float64; returns rescaled embeddings given masked embeddings p_emb and projectors."""
    def run(P_m, P_c):
        n_t, dim = p_emb.shape
        I_c = np.eye(dim) - P_c
        dist = np.linalg.norm(I_c @ p_emb.T, axis=0)
        means = np.array([np.mean(np.delete(dist, i)) for i in range(n_t)])
        rm = (dist < (1.0 + alpha) * means).astype(np.float64)
        ones = np.ones(max_length); ones[1:n_t + 1] = rm
        text = ie[1]
        new = (I_c @ P_m @ text.T).T
        merged = np.where(ones[:, None].astype(bool), text, new)
        return np.stack([ie[0], merged]), int(n_t - rm.sum())
    return run


def f_beta(z, btype="sigmoid", upperbound_timestep=10, concept_type="nudity"):
    """Self-validation filter step count (...threshold_time.py:19-36); pinned by the f_beta grid of safree_golden.npz."""
    t, k = (5.5, 3.5) if "artists-" in concept_type else (5.333, 2.5)
    if btype == "tanh":
        return round(upperbound_timestep / 2.0 * (np.tanh(k * (10 * z - t)) + 1))
    if btype == "sigmoid":
        return round(upperbound_timestep * (1.0 / (1.0 + np.exp(-2.0 * k * (10 * z - t)))))
    raise NotImplementedError("btype is incorrect")


def prepare(pair, masked, negspace, attention_mask, alpha=0.01, up_t=10, category="nudity"):
    """The SAFREE block of the reference's __call__ for ONE prompt (...threshold_time.py:458-486) in float64 numpy: concept
    projector, masked-token projector, trigger-token test + token-wise replacement, and the self-validation statistic
    beta = 1 - mean cos(text, (I - P_c) P_m text) over the attended positions -> f_beta.  pair [2,77,dim] = (uncond, text);
    masked [n_t,dim]; negspace [n_neg,dim]; attention_mask [77].  Returns rescaled pair, token keep-mask [77] (True = kept),
    n_removed, beta, beta_adjusted."""
    pair, masked, negspace = (np.asarray(a, dtype=np.float64) for a in (pair, masked, negspace))
    P_c, P_m = proj(negspace.T), proj(masked.T)
    n_t, dim = masked.shape
    I_c = np.eye(dim) - P_c
    dist = np.linalg.norm(I_c @ masked.T, axis=0)
    means = np.array([np.mean(np.delete(dist, i)) for i in range(n_t)]) if n_t > 1 else np.full(n_t, np.nan)
    keep = dist < (1.0 + alpha) * means
    mask = np.ones(pair.shape[1], dtype=bool)
    mask[1:n_t + 1] = keep
    text = pair[1]
    ort = (I_c @ P_m @ text.T).T
    rescaled = np.stack([pair[0], np.where(mask[:, None], text, ort)])
    act = np.asarray(attention_mask).reshape(-1) == 1
    cos = np.sum(ort[act] * text[act], -1) / np.maximum(np.linalg.norm(ort[act], axis=-1) * np.linalg.norm(text[act], axis=-1), 1e-8)
    beta = 1.0 - float(cos.mean())
    return {"rescaled": rescaled, "mask": mask, "n_removed": int(n_t - keep.sum()), "beta": beta,
            "beta_adjusted": f_beta(beta, upperbound_timestep=up_t, concept_type=category),
            "margin": float(np.min(np.abs(dist - (1.0 + alpha) * means) / np.maximum(dist, 1e-30))) if n_t > 1 else float("nan")}
