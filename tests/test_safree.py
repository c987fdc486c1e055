"""SAFREE text projection (next-row f1): torch implementation vs an independent numpy restatement; f_beta KATs."""
import numpy as np
import torch

from oracle import safree as osf
from safe_denoiser_amd import safree


def test_f_beta_known_values():
    # sigmoid(2*2.5*(10 z - 5.333)) * 10, rounded
    assert safree.f_beta(0.0) == 0 and safree.f_beta(1.0) == 10
    assert safree.f_beta(0.5333) == 5
    assert safree.f_beta(0.6, btype="tanh") == round(5 * (np.tanh(2.5 * (6 - 5.333)) + 1))
    assert safree.f_beta(0.55, concept_type="artists-VanGogh") == round(10 / (1 + np.exp(-2 * 3.5 * (5.5 - 5.5))))


def test_projection_and_token_replacement_match_numpy():
    g = torch.Generator().manual_seed(0)
    dim, n_t, n_neg = 96, 9, 17
    E = torch.randn(2, 77, dim, generator=g, dtype=torch.float64)
    neg = torch.randn(n_neg, dim, generator=g, dtype=torch.float64)
    masked = torch.randn(n_t, dim, generator=g, dtype=torch.float64)
    masked[3] = neg[:4].mean(0) * 3 + 0.05 * masked[3]                      # token 3 lies near the concept space
    am = torch.zeros(77); am[:n_t + 2] = 1
    out = safree.prepare(E, masked, neg, am, alpha=0.01, svf=True, up_t=10)
    P_c, P_m = osf.proj(neg.numpy().T), osf.proj(masked.numpy().T)
    # projector properties
    np.testing.assert_allclose(safree.projection_matrix(neg.T).numpy(), P_c, atol=1e-9)
    np.testing.assert_allclose(P_c @ P_c, P_c, atol=1e-9)
    ref, n_removed = osf.safree(E.numpy(), masked.numpy(), 0.01)(P_m, P_c)
    np.testing.assert_allclose(out["rescaled_text_embeddings"].numpy(), ref, atol=1e-9)
    assert out["n_removed"] == n_removed >= 1
    # untouched rows: unconditional branch, BOS token, padding beyond the prompt
    assert torch.equal(out["rescaled_text_embeddings"][0], E[0])
    assert torch.equal(out["rescaled_text_embeddings"][1, 0], E[1, 0])
    assert torch.equal(out["rescaled_text_embeddings"][1, n_t + 1:], E[1, n_t + 1:])
    assert 0 <= out["beta_adjusted"] <= 10 and 0.0 <= out["beta"] <= 2.0


# ---- pinned against the reference's own helpers (tests/golden/make_safree_golden.py: the functions at
# ...threshold_time.py:16-99 executed in the build container; inputs/outputs only) ----
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "safree_golden.npz")


def _gold():
    z = np.load(GOLD)
    return z, [str(n) for n in z["__cases__"]]


def test_f_beta_matches_reference_grid():
    z, _ = _gold()
    rows = iter(z["f_beta/out"])
    for btype in ("sigmoid", "tanh"):
        for concept in ("nudity", "artists-VanGogh"):
            for up_t in (10, 20):
                want = next(rows)
                got = [safree.f_beta(float(v), btype=btype, upperbound_timestep=up_t, concept_type=concept) for v in z["f_beta/z"]]
                assert got == list(want), (btype, concept, up_t)


def test_projection_helpers_match_reference_goldens():
    z, names = _gold()
    for name in names:
        ie, neg, p_emb = (torch.from_numpy(z[f"{name}/{k}"]) for k in ("ie", "neg", "p_emb"))
        alpha = float(z[f"{name}/alpha"])
        tol = 1e-9 if ie.dtype == torch.float64 else 2e-4            # fp32 pinverse of a 17x17 / n_t x n_t Gram matrix
        P_c, P_m = safree.projection_matrix(neg.T), safree.projection_matrix(p_emb.T)
        if f"{name}/P_c" in z.files:
            np.testing.assert_allclose(P_c.numpy(), z[f"{name}/P_c"], atol=tol)
            np.testing.assert_allclose(P_m.numpy(), z[f"{name}/P_m"], atol=tol)
            # the numpy oracle agrees with the reference too
            np.testing.assert_allclose(osf.proj(neg.numpy().T), z[f"{name}/P_c"], atol=tol)
        resc, n_removed = safree.safree_projection(ie, p_emb, P_m, P_c, alpha=alpha, max_length=77)
        assert n_removed == int(z[f"{name}/n_removed"]), name
        np.testing.assert_allclose(resc.numpy(), z[f"{name}/rescaled"], atol=tol * 10, rtol=tol * 10)
        ort = safree.projection_and_orthogonal(ie, P_m, P_c)
        np.testing.assert_allclose(ort.numpy(), z[f"{name}/proj_ort"], atol=tol * 10, rtol=tol * 10)
        if ie.dtype == torch.float64:
            ref, nr = osf.safree(ie.numpy(), p_emb.numpy(), alpha)(osf.proj(p_emb.numpy().T), osf.proj(neg.numpy().T))
            assert nr == n_removed
            np.testing.assert_allclose(ref, z[f"{name}/rescaled"], atol=1e-9)


def test_sd3_mask_to_onp_matches_reference_goldens():
    """models/sdv3/safe_denoiser_pipeline.py:72-153 (fp32 projectors, bfloat16 products, 333-token axis)."""
    z, _ = _gold()
    for name in [str(n) for n in z["__sd3_cases__"]]:
        dt = torch.float16 if int(z[f"{name}/f16"]) else torch.float32
        ie, neg, p_emb = (torch.from_numpy(z[f"{name}/{k}"]).to(dt) for k in ("ie", "neg", "p_emb"))
        out = safree.prepare_sd3(ie, p_emb, neg, alpha=float(z[f"{name}/alpha"]))
        assert out["n_removed"] == float(z[f"{name}/n_removed"]), name
        np.testing.assert_array_equal(out["sp_vector"].numpy(), z[f"{name}/keep"])
        np.testing.assert_array_equal(out["inv_vector"].numpy(), z[f"{name}/inv"])
        np.testing.assert_allclose(out["rescaled_text_embeddings"].float().numpy(), z[f"{name}/rescaled"], atol=2e-2, rtol=2e-2)
        assert out["rescaled_text_embeddings"].shape == (2, 333, ie.shape[-1])


def test_oracle_prepare_and_f_beta_are_pinned_to_the_reference_goldens():
    """oracle.safree.prepare / f_beta (the float64 truth of tests/test_gpu_e2e_ids.py) against the reference's own helpers: the
    f_beta grid exactly; rescaled embeddings, removed-token count and the projected text to the goldens' precision; the product's
    per-prompt and batched paths report the same token mask."""
    z, names = _gold()
    rows = iter(z["f_beta/out"])
    for btype in ("sigmoid", "tanh"):
        for concept in ("nudity", "artists-VanGogh"):
            for up_t in (10, 20):
                want = next(rows)
                assert [osf.f_beta(float(v), btype=btype, upperbound_timestep=up_t, concept_type=concept) for v in z["f_beta/z"]] == list(want)
    for name in names:
        ie, neg, p_emb = (z[f"{name}/{k}"] for k in ("ie", "neg", "p_emb"))
        tol = 1e-9 if ie.dtype == np.float64 else 2e-3
        am = np.zeros(77); am[:p_emb.shape[0] + 2] = 1
        o = osf.prepare(ie, p_emb, neg, am, alpha=float(z[f"{name}/alpha"]))
        assert o["n_removed"] == int(z[f"{name}/n_removed"]), name
        np.testing.assert_allclose(o["rescaled"], z[f"{name}/rescaled"], atol=tol, rtol=tol)
        assert o["mask"].sum() == 77 - o["n_removed"]
        t = torch.from_numpy(ie).double()
        pr = safree.prepare(t, torch.from_numpy(p_emb).double(), torch.from_numpy(neg).double(), torch.from_numpy(am), alpha=float(z[f"{name}/alpha"]))
        assert np.array_equal(pr["token_mask"].numpy(), o["mask"]) and pr["beta_adjusted"] == o["beta_adjusted"]
        assert abs(pr["beta"] - o["beta"]) <= 1e-9
        pb = safree.prepare_batch(torch.cat([t[:1], t[:1], t[1:], t[1:]]), [torch.from_numpy(p_emb).double()] * 2, torch.from_numpy(neg).double(),
                                  torch.from_numpy(np.stack([am, am])), alpha=float(z[f"{name}/alpha"]))
        assert np.array_equal(pb["token_mask"][1].numpy(), o["mask"]) and pb["beta_adjusted"] == [o["beta_adjusted"]] * 2
