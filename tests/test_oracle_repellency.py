"""Oracle (CPU restatement) vs golden vectors captured from the reference's own modules.

Tolerance: the oracle uses the same torch CPU ops as the reference, so fp32 results agree to a few ulp;
rtol=2e-6 / atol=1e-7 is asserted.
"""
import os

import numpy as np
import pytest
import torch

from oracle import repellency as orp

RT, AT = 2e-6, 1e-7


def T(a):
    return torch.from_numpy(np.array(a)).clone()


def close(a, b, rt=RT, at=AT):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rt, atol=at, equal_nan=True)


def chan_norm_refs(m, c, hw, seed):
    g = torch.Generator().manual_seed(seed)
    r = torch.randn(m, c, hw, hw, generator=g)
    return r / torch.norm(r, dim=1, keepdim=True)


def test_case_inventory(golden):
    for prefix in ("G1_", "G2_", "G3a_", "G3b_", "G3c_", "G4_", "G5_", "G6_beta", "G6_radius", "G7_", "G8_"):
        assert any(k.startswith(prefix) for k in golden), prefix


def test_g1_threshold_gate(golden):
    n = 0
    for name, c in golden.items():
        if not name.startswith("G1_"):
            continue
        x = T(c["x"])
        out = orp.kernel_fast_conditioning(x, T(c["refs"]), flavour="threshold", scale=float(c["scale"]),
                                           epsilon=float(c["epsilon"]), sigma=float(c["sigma"]),
                                           beta_threshold=float(c["beta_threshold"]),
                                           beta_threshold_margin=float(c["margin"]), use_beta_threshold=True)
        assert out["x_0_hat"] is x                                  # in-place aliasing
        close(out["x_0_hat"], c["out_x"])
        close(out["mean_x_0_hat"]["denominator"], c["den"])
        assert int(out["is_negation"]) == int(c["isneg"])
        n += 1
    assert n >= 24
    # both branches of the gate are exercised
    assert {int(c["isneg"]) for k, c in golden.items() if k.startswith("G1_")} == {0, 1}


def test_g2_threshold_returns_negative_score(golden):
    for name, c in golden.items():
        if not name.startswith("G2_"):
            continue
        x = T(c["x"])
        out = orp.kernel_fast_conditioning(x, T(c["refs"]), flavour="threshold", scale=float(c["scale"]),
                                           epsilon=float(c["epsilon"]), sigma=float(c["sigma"]),
                                           use_beta_threshold=False)
        close(out["x_0_hat"], c["out_neg"])
        close(x, c["mutated_x"])
        assert out["is_negation"] is True and int(c["isneg"]) == 1


def test_g3_fast_variants(golden):
    for name, c in golden.items():
        if name.startswith("G3a_"):
            x = T(c["x"])
            out = orp.kernel_fast_conditioning(x, T(c["refs"]), flavour="fast", scale=float(c["scale"]),
                                               epsilon=float(c["epsilon"]), sigma=3.15)   # sigma must be ignored
            assert out["x_0_hat"] is x and int(c["alias"]) == 1
            assert "is_negation" not in out
            close(out["x_0_hat"], c["out_x"])
        elif name.startswith("G3b_"):
            x = T(c["x"])
            out = orp.kernel_fast_conditioning(x, T(c["refs"]), flavour="fast", scale=float(c["scale"]),
                                               epsilon=float(c["epsilon"]), guidance_scale=1.0)
            close(out["x_0_hat"], c["out_neg"])
            close(x, c["mutated_x"])
        elif name.startswith("G3c_"):
            xh = T(c["x"]).half()
            keep = xh.clone()
            out = orp.kernel_fast_conditioning(xh, T(c["refs"]), flavour="fast", scale=float(c["scale"]),
                                               epsilon=float(c["epsilon"]))
            assert out["x_0_hat"].dtype == torch.float32 and int(c["out_is_f32"]) == 1
            assert torch.equal(xh, keep) and int(c["input_unchanged"]) == 1
            close(out["x_0_hat"], c["out_x"])


def test_g4_sdv3_query_norm_and_nan(golden):
    for name, c in golden.items():
        if not name.startswith("G4_"):
            continue
        x = T(c["x"])
        out = orp.kernel_fast_conditioning(x, T(c["refs"]), flavour="fast_sdv3", scale=float(c["scale"]),
                                           epsilon=float(c["epsilon"]))
        close(out["x_0_hat"], c["out_x"])
    assert np.isnan(golden["G4_nan"]["out_x"]).all()                # NaN floods the whole output


def test_g5_sparse(golden):
    for name, c in golden.items():
        if not name.startswith("G5_"):
            continue
        flavour = name[len("G5_"):].rsplit("_", 1)[0]
        x = T(c["x"])
        out = orp.sparse_conditioning(x, T(c["refs"]), flavour=flavour, radius=float(c["radius"]),
                                      scale=float(c["scale"]))
        close(out["x_0_hat"], c["out_x"], rt=1e-5, at=1e-6)
        close(out["mean_x_0_hat"], c["force_norm"], rt=1e-5, at=1e-6)
        assert int(out.get("is_negation", -1)) == int(c["isneg"])
    assert int(golden["G5_threshold_none"]["isneg"]) == 0 and int(golden["G5_threshold_some"]["isneg"]) == 1


def test_g6_calibration(golden):
    for name, c in golden.items():
        if name.startswith("G6_beta"):
            noisy = {981: T(c["noisy981"]), 1: T(c["noisy1"])}
            res = orp.empirical_beta(noisy, T(c["refs"]), float(c["sigma"]), float(c["epsilon"]), float(c["q"]))
            close(res[981], c["beta981"]); close(res[1], c["beta1"])
            close(orp.calibrated_threshold(res), c["beta1"])
        elif name.startswith("G6_radius"):
            noisy = {981: T(c["noisy981"]), 1: T(c["noisy1"])}
            res = orp.empirical_radius(noisy, T(c["refs"]), float(c["q"]))
            close(res[981], c["radius981"]); close(res[1], c["radius1"])


def test_g7_project_chunking(golden):
    def fake_embed(img):
        z = torch.nn.functional.avg_pool2d(img, 8)
        return torch.cat([z, z.sum(1, keepdim=True)], 1)
    for name, c in golden.items():
        if not name.startswith("G7_"):
            continue
        flavour = name.split("_")[1]
        out = orp.project_refs(T(c["imgs"]), fake_embed, int(c["n_embed"]), flavour)
        close(out, c["out"])
        close(torch.linalg.vector_norm(out, dim=1), np.ones_like(c["out"][:, 0]), rt=1e-5)


@pytest.mark.parametrize("case", ["G8_fast_full", "G8_threshold_full"])
def test_g8_full_size(golden, case):
    c = golden[case]
    refs = chan_norm_refs(int(c["m"]), 4, 64, int(c["seed_refs"]))
    x = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(int(c["seed_x"])))
    x0 = x.clone()
    if case == "G8_fast_full":
        out = orp.kernel_fast_conditioning(x, refs, flavour="fast", scale=float(c["scale"]), epsilon=float(c["epsilon"]))
        # sigma=1, d~143: every weight underflows to ~1e-31, epsilon dominates -> the update is ~0
        assert float((out["x_0_hat"] - x0).abs().max()) <= 1e-6 and float(c["max_abs_delta"]) <= 1e-6
    else:
        out = orp.kernel_fast_conditioning(x, refs, flavour="threshold", scale=float(c["scale"]),
                                           epsilon=float(c["epsilon"]), sigma=float(c["sigma"]),
                                           beta_threshold=float(c["beta_threshold"]),
                                           beta_threshold_margin=float(c["margin"]), use_beta_threshold=True)
        close(out["mean_x_0_hat"]["denominator"], c["den"], rt=1e-5)
        assert int(out["is_negation"]) == int(c["isneg"])
    ox = out["x_0_hat"].reshape(-1)
    close(ox[:16], c["head"], rt=1e-5); close(ox[-16:], c["tail"], rt=1e-5)
    close(float(ox.double().sum()), c["sum64"], rt=1e-6, at=1e-3)
    close(float(ox.double().norm()), c["l2_64"], rt=1e-6)


def sdv3_full_case(tag, refs, seed_noise=2000):
    """Inputs of tests/golden/repellency_golden_sdv3_full.npz (make_golden.py --sdv3-full), rebuilt from their seeds."""
    noise = torch.randn(1, 16, 64, 64, generator=torch.Generator().manual_seed(seed_noise))
    if tag in ("near", "near_f16"):
        x = 3.0 * (refs[7:8].clone() + 0.0005 * noise)
        return x.half() if tag == "near_f16" else x
    if tag == "between":
        return 2.0 * (0.5 * refs[100:101] + 0.5 * refs[300:301] + 0.0003 * noise)
    return noise.clone()


SDV3_FULL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "repellency_golden_sdv3_full.npz")


@pytest.mark.parametrize("tag", ["near", "between", "far", "near_f16"])
def test_g8_sdv3_full_size(tag):
    """BASELINE config 4's projection at its real size (M = 515 references of [16, 64, 64], D = 65 536) against the reference's
    own repellency_methods_fast_sdv3 (VERDICT r3 missing #4): head / tail / float64 sum / L2 of the output and the size of the
    update.  `near`: weights of order 1; `between`: a 7e-5 update (cancellation-limited: 2 % on its norm); `far`: every weight
    underflows against epsilon, output = input; `near_f16`: the fp16 latents the SD-v3 pipelines hand over."""
    z = np.load(SDV3_FULL)
    c = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(f"G8_sdv3_full_{tag}/")}
    refs = chan_norm_refs(int(c["m"]), 16, 64, int(c["seed_refs"]))
    x = sdv3_full_case(tag, refs, int(c["seed_noise"]))
    out = orp.kernel_fast_conditioning(x.clone(), refs, flavour="fast_sdv3", scale=float(c["scale"]), epsilon=float(c["epsilon"]))["x_0_hat"]
    assert out.dtype == torch.float32 and int(c["out_is_f32"]) == 1
    ox = out.reshape(-1)
    close(ox[:16], c["head"], rt=1e-5); close(ox[-16:], c["tail"], rt=1e-5)
    close(float(ox.double().sum()), c["sum64"], rt=1e-6, at=1e-3)
    close(float(ox.double().norm()), c["l2_64"], rt=1e-6)
    d = float((out.double() - x.double()).norm())
    assert abs(d - float(c["delta_l2"])) <= 2e-2 * float(c["delta_l2"]) + 1e-9
    if tag == "far":
        assert d == 0.0
