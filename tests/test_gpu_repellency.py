"""HIP repellency path (through the C ABI) vs the golden vectors captured from the reference and vs the oracle.

Tolerance: fp32 arithmetic with a different summation order than torch (direct-difference distance, sliced
weighted sum) -> rtol 2e-5 / atol 2e-6 on latents; gates (is_negation) must match exactly.
"""
import numpy as np
import pytest
import torch

from oracle import repellency as orp

pytestmark = pytest.mark.gpu
RT, AT = 2e-5, 2e-6


def T(a):
    return torch.from_numpy(np.array(a)).clone()


def close(a, b, rt=RT, at=AT):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rt, atol=at, equal_nan=True)


def make_proc(mod, method, refs, tmp_path, **params):
    path = str(tmp_path / f"pr_{abs(hash((method, str(params)))) % 10**9}.pt")
    torch.save(refs, path)
    ref_data = torch.zeros(1, dtype=torch.float32, device="cuda")
    return mod.get_repellency_method(method, ref_data, None, None, 50, 1000, 0.00085, 0.012, n_embed=4,
                                     proj_ref_path=path, cache_proj_ref=True, **params)


def test_library_sees_gfx950():
    import safe_denoiser_amd as sda
    assert sda.lib().sdn_device_arch_host() == b"gfx950"


def test_g1_g2_threshold(golden, tmp_path):
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
    n1 = n2 = 0
    for name, c in golden.items():
        if name.startswith("G1_"):
            proc = make_proc(thr, "kernel_fast", T(c["refs"]), tmp_path, sigma=float(c["sigma"]),
                             scale=float(c["scale"]), beta_threshold=float(c["beta_threshold"]),
                             beta_threshold_margin=float(c["margin"]))
            x = T(c["x"]).cuda()
            out = proc.conditioning(x, beta_threshold=True)
            assert out["x_0_hat"] is x
            close(out["x_0_hat"], c["out_x"])
            close(out["mean_x_0_hat"]["denominator"], c["den"], rt=1e-5)
            assert int(out["is_negation"]) == int(c["isneg"]), name
            close(out["mean_x_0_hat"]["negative_score_item"], c["item"], rt=1e-4, at=1e-6)
            n1 += 1
        elif name.startswith("G2_"):
            proc = make_proc(thr, "kernel_fast", T(c["refs"]), tmp_path, sigma=float(c["sigma"]),
                             scale=float(c["scale"]), beta_threshold=1.0)
            x = T(c["x"]).cuda()
            out = proc.conditioning(x, beta_threshold=False)
            close(out["x_0_hat"], c["out_neg"])
            close(x, c["mutated_x"])
            assert out["is_negation"] is True
            n2 += 1
    assert n1 >= 24 and n2 >= 12


def test_g3_fast(golden, tmp_path):
    from safe_denoiser_amd.repellency import repellency_methods_fast as fast
    for name, c in golden.items():
        if name.startswith("G3a_"):
            proc = make_proc(fast, "kernel_fast", T(c["refs"]), tmp_path, sigma=3.15, scale=float(c["scale"]))
            x = T(c["x"]).cuda()
            out = proc.conditioning(x)
            assert out["x_0_hat"] is x and "is_negation" not in out
            close(out["x_0_hat"], c["out_x"])
        elif name.startswith("G3b_"):
            proc = make_proc(fast, "kernel_fast", T(c["refs"]), tmp_path, scale=float(c["scale"]))
            x = T(c["x"]).cuda()
            out = proc.conditioning(x, guidance_scale=1.0)
            close(out["x_0_hat"], c["out_neg"])
            close(x, c["mutated_x"])
        elif name.startswith("G3c_"):
            proc = make_proc(fast, "kernel_fast", T(c["refs"]), tmp_path, scale=float(c["scale"]))
            xh = T(c["x"]).half().cuda()
            keep = xh.clone()
            out = proc.conditioning(xh)
            assert out["x_0_hat"].dtype == torch.float32 and torch.equal(xh, keep)
            close(out["x_0_hat"], c["out_x"])


def test_g4_sdv3(golden, tmp_path):
    from safe_denoiser_amd.repellency import repellency_methods_fast_sdv3 as sd3
    for name, c in golden.items():
        if not name.startswith("G4_"):
            continue
        proc = make_proc(sd3, "kernel_fast", T(c["refs"]), tmp_path, scale=float(c["scale"]))
        x = T(c["x"]).cuda()
        out = proc.conditioning(x)
        close(out["x_0_hat"], c["out_x"])
        if name == "G4_nan":
            assert torch.isnan(out["x_0_hat"]).all()


def test_g5_sparse(golden, tmp_path):
    from safe_denoiser_amd.repellency import (repellency_methods_fast, repellency_methods_fast_sdv3,
                                              repellency_methods_threshold)
    mods = {"threshold": repellency_methods_threshold, "fast": repellency_methods_fast,
            "fast_sdv3": repellency_methods_fast_sdv3}
    for name, c in golden.items():
        if not name.startswith("G5_"):
            continue
        flavour = name[len("G5_"):].rsplit("_", 1)[0]
        proc = make_proc(mods[flavour], "sparse", T(c["refs"]), tmp_path, radius=float(c["radius"]),
                         scale=float(c["scale"]))
        x = T(c["x"]).cuda()
        out = proc.conditioning(x, beta_threshold=True) if flavour == "threshold" else proc.conditioning(x)
        close(out["x_0_hat"], c["out_x"], rt=5e-5, at=5e-5)
        close(out["mean_x_0_hat"], c["force_norm"], rt=1e-4, at=1e-4)
        assert int(out.get("is_negation", -1)) == int(c["isneg"]), name


def test_g6_calibration(golden, tmp_path):
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
    for name, c in golden.items():
        noisy = None
        if name.startswith("G6_"):
            noisy = {981: T(c["noisy981"]).cuda(), 1: T(c["noisy1"]).cuda()}
        if name.startswith("G6_beta"):
            proc = make_proc(thr, "kernel_fast", T(c["refs"]), tmp_path, sigma=3.15, beta_threshold=1.0)
            proc.noisy_proj_refs = noisy
            res = proc.empirical_beta(sigma=float(c["sigma"]), quantitle=float(c["q"]))
            close(res[981], c["beta981"], rt=1e-5); close(res[1], c["beta1"], rt=1e-5)
        elif name.startswith("G6_radius"):
            proc = make_proc(thr, "sparse", T(c["refs"]), tmp_path, radius=1.0)
            proc.noisy_proj_refs = noisy
            res = proc.empirical_radius(quantitle=float(c["q"]))
            close(res[981], c["radius981"], rt=1e-5); close(res[1], c["radius1"], rt=1e-5, at=1e-5)


def test_auto_calibration_uses_last_timestep(tmp_path):
    """beta_threshold <= 0 -> noisy refs for every timestep with the seed-42 device generator, empirical_beta,
    keep the LAST key (t = 1).  Compared with the oracle fed the same noisy refs."""
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
    from safe_denoiser_amd.schedulers import DDPMScheduler
    g = torch.Generator().manual_seed(0)
    refs = orp.channel_normalise(torch.randn(12, 4, 8, 8, generator=g))
    sch = DDPMScheduler()
    proc = make_proc(thr, "kernel_fast", refs, tmp_path, sigma=3.15, scale=0.33, quantile=0.25, scheduler=sch,
                     proj_noisy_ref_path_for_beta=str(tmp_path / "noisy.pt"))
    noisy = torch.load(str(tmp_path / "noisy.pt"), map_location="cpu")
    assert list(noisy.keys()) == list(range(981, 0, -20))
    exp = orp.calibrated_threshold(orp.empirical_beta(noisy, refs, 3.15, 1e-8, 0.25))
    close(proc.beta_threshold, exp, rt=1e-5)


@pytest.mark.parametrize("case", ["G8_fast_full", "G8_threshold_full"])
def test_g8_full_size(golden, tmp_path, case):
    from safe_denoiser_amd.repellency import repellency_methods_fast as fast
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
    c = golden[case]
    g = torch.Generator().manual_seed(int(c["seed_refs"]))
    refs = orp.channel_normalise(torch.randn(int(c["m"]), 4, 64, 64, generator=g))
    x = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(int(c["seed_x"]))).cuda()
    if case == "G8_fast_full":
        proc = make_proc(fast, "kernel_fast", refs, tmp_path, scale=float(c["scale"]))
        out = proc.conditioning(x)
    else:
        proc = make_proc(thr, "kernel_fast", refs, tmp_path, sigma=float(c["sigma"]), scale=float(c["scale"]),
                         beta_threshold=float(c["beta_threshold"]), beta_threshold_margin=float(c["margin"]))
        out = proc.conditioning(x, beta_threshold=True)
        close(out["mean_x_0_hat"]["denominator"], c["den"], rt=2e-5)
        assert int(out["is_negation"]) == int(c["isneg"])
    ox = out["x_0_hat"].reshape(-1)
    close(ox[:16], c["head"], rt=2e-5, at=2e-6); close(ox[-16:], c["tail"], rt=2e-5, at=2e-6)
    close(float(ox.double().sum()), c["sum64"], rt=1e-5, at=2e-3)
    close(float(ox.double().norm()), c["l2_64"], rt=1e-6)


@pytest.mark.parametrize("n", [2, 8, 11, 64])
def test_batched_queries_equal_per_sample_oracle(tmp_path, n):
    """N > 1 (the engine's batching; the reference is fixed at N = 1): every row must equal the N = 1 oracle."""
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
    g = torch.Generator().manual_seed(5)
    refs = orp.channel_normalise(torch.randn(37, 4, 16, 16, generator=g))
    x = torch.randn(n, 4, 16, 16, generator=g)
    x[1] = refs[3] * 1.1                                          # one query close to a reference
    proc = make_proc(thr, "kernel_fast", refs, tmp_path, sigma=3.15, scale=0.33, beta_threshold=3.0,
                     beta_threshold_margin=1.6)
    xg = x.clone().cuda()
    _neg, den, isneg = proc.conditioning_device(xg, beta_threshold=True)
    for i in range(n):
        xi = x[i:i + 1].clone()
        o = orp.kernel_fast_conditioning(xi, refs, flavour="threshold", scale=0.33, sigma=3.15, beta_threshold=3.0,
                                         beta_threshold_margin=1.6, use_beta_threshold=True)
        close(xg[i:i + 1], o["x_0_hat"])
        close(den[i], o["mean_x_0_hat"]["denominator"], rt=1e-5)
        assert bool(isneg[i].item()) == o["is_negation"]


def test_edge_cases(tmp_path):
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
    # M = 1, D = 4 (smallest legal), and a query identical to the reference (distance exactly 0)
    refs = orp.channel_normalise(torch.tensor([[[[1.0]], [[2.0]], [[-1.0]], [[0.5]]]]))
    proc = make_proc(thr, "kernel_fast", refs, tmp_path, sigma=3.15, scale=0.33, beta_threshold=0.5)
    x = refs.clone().cuda()
    out = proc.conditioning(x, beta_threshold=True)
    o = orp.kernel_fast_conditioning(refs.clone(), refs, flavour="threshold", scale=0.33, sigma=3.15,
                                     beta_threshold=0.5, use_beta_threshold=True)
    close(out["x_0_hat"], o["x_0_hat"])
    assert out["is_negation"] is True and o["is_negation"] is True
    # shape mismatch is an error, not a silent reshape
    import safe_denoiser_amd as sda
    with pytest.raises(sda.SdnError):
        proc.conditioning(torch.randn(1, 4, 2, 2, device="cuda"), beta_threshold=True)


def test_g7_goldens_against_product_project(golden, tmp_path):
    """Row R6: the product's RepellencyMethod.project() (chunks of n_embed only when len > n_embed, per-pixel channel
    normalisation, the fast flavour's float cast) on the GPU vs the outputs the reference's project() gave for the same
    images and the same fake embed_fn (tests/golden/make_golden.py G7)."""
    from safe_denoiser_amd.repellency import repellency_methods_fast as fast
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr

    calls = []

    def fake_embed(img):
        calls.append(int(img.shape[0]))
        z = torch.nn.functional.avg_pool2d(img, 8)
        return torch.cat([z, z.sum(1, keepdim=True)], 1)

    n = 0
    for name, c in golden.items():
        if not name.startswith("G7_"):
            continue
        flavour = name.split("_")[1]
        refs = orp.channel_normalise(torch.randn(3, 4, 4, 4, generator=torch.Generator().manual_seed(600)))
        proc = make_proc(thr if flavour == "threshold" else fast, "kernel_fast", refs, tmp_path, beta_threshold=1.0)
        proc.embed_fn = fake_embed
        proc.n_embed = int(c["n_embed"])
        calls.clear()
        imgs = T(c["imgs"]).cuda()
        out = proc.project(imgs)
        assert out.is_cuda and out.dtype == torch.float32
        close(out, c["out"])
        k, ne = imgs.shape[0], int(c["n_embed"])
        assert calls == ([k] if k <= ne else [min(ne, k - i) for i in range(0, k, ne)]), (name, calls)
        n += 1
    assert n == 6


@pytest.mark.parametrize("tag", ["near", "between", "far", "near_f16"])
def test_g8_sdv3_full_size_on_the_hip_path(tmp_path, tag):
    """BASELINE config 4's projection at full size (M = 515, C = 16, 64 x 64: D = 65 536) through the product's
    repellency_methods_fast_sdv3 front end vs the goldens captured from the reference's module (make_golden.py --sdv3-full)."""
    from safe_denoiser_amd.repellency import repellency_methods_fast_sdv3 as sd3rep
    from tests.test_oracle_repellency import SDV3_FULL, chan_norm_refs, sdv3_full_case
    z = np.load(SDV3_FULL)
    c = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(f"G8_sdv3_full_{tag}/")}
    refs = chan_norm_refs(int(c["m"]), 16, 64, int(c["seed_refs"]))
    x = sdv3_full_case(tag, refs, int(c["seed_noise"]))
    proc = make_proc(sd3rep, "kernel_fast", refs, tmp_path, scale=float(c["scale"]))
    out = proc.conditioning(x.clone().cuda())["x_0_hat"]
    assert out.dtype == torch.float32
    ox = out.reshape(-1)
    close(ox[:16], c["head"], rt=2e-5, at=2e-6); close(ox[-16:], c["tail"], rt=2e-5, at=2e-6)
    close(float(ox.double().sum()), c["sum64"], rt=1e-5, at=5e-3)
    close(float(ox.double().norm()), c["l2_64"], rt=1e-6)
    d = float((out.double().cpu() - x.double()).norm())
    assert abs(d - float(c["delta_l2"])) <= 3e-2 * float(c["delta_l2"]) + 1e-9, (d, float(c["delta_l2"]))


def test_fast_sdv3_at_1024_squared_against_the_oracle(tmp_path):
    """BASELINE config 4 names 1024 x 1024: latents [16, 128, 128], D = 262 144, M = 515 (540 MB of references; the reference's
    driver runs 512 x 512 and ships no 1024-sized cache).  HIP vs the CPU oracle on a query next to a reference, one
    between two, a batch of three -- and the size-independent property that a query which IS alpha x reference k (any alpha > 0:
    the query is channel-normalised) comes back as x - scale (r_k + sum of the others' weighted rows) / (1 + ...)."""
    from safe_denoiser_amd.repellency import repellency_methods_fast_sdv3 as sd3rep
    g = torch.Generator().manual_seed(3)
    refs = orp.channel_normalise(torch.randn(515, 16, 128, 128, generator=g))
    noise = torch.randn(3, 16, 128, 128, generator=g)
    x = torch.cat([2.5 * (refs[11:12] + 0.0003 * noise[:1]), 0.7 * (0.5 * refs[1:2] + 0.5 * refs[2:3] + 0.0002 * noise[1:2]), noise[2:3]])
    proc = make_proc(sd3rep, "kernel_fast", refs, tmp_path, scale=0.03)
    xg = x.clone().cuda()
    neg, den, isneg = proc.conditioning_device(xg)
    for p in range(3):
        want = orp.kernel_fast_conditioning(x[p:p + 1].clone(), refs, flavour="fast_sdv3", scale=0.03)["x_0_hat"]
        close(xg[p:p + 1], want, rt=2e-5, at=2e-6)
    assert float((xg[2].cpu() - x[2]).abs().max()) == 0.0                     # epsilon-dominated: untouched
    for alpha in (0.1, 7.0):
        q = (alpha * refs[40:41]).clone().cuda()
        q0 = q.clone()
        proc.conditioning_device(q)
        upd = (q0 - q).cpu() / 0.03                                           # = neg = sum_m w_m r_m / (sum_m w_m + eps)
        # w_40 = exp(0) = 1; every other reference is ~ sqrt(2) * 128 away: weight exp(-90) -> neg = r_40 / (1 + 1e-8)
        close(upd, refs[40:41], rt=0, at=1e-6 + 4e-7 * alpha / 0.03)         # (q0 - q is formed in fp32 at the query's magnitude)
