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
