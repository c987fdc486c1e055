"""Size-independent properties at BASELINE.json's FULL sizes (M = 515 references, D = 4*64*64, SD-v1.4 UNet shapes),
where a CPU oracle run per case would be too slow, plus the maximum / empty reference-set edge cases."""
import ctypes as C

import pytest
import torch

import safe_denoiser_amd as sda
from oracle import repellency as orp
from safe_denoiser_amd import _lib
from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
from safe_denoiser_amd.schedulers import DDIMScheduler, DDPMScheduler

pytestmark = pytest.mark.gpu


def make_proc(refs, tmp_path, method="kernel_fast", **params):
    path = str(tmp_path / f"pr_{abs(hash((method, str(params), refs.shape[0]))) % 10**9}.pt")
    torch.save(refs, path)
    return thr.get_repellency_method(method, torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085, 0.012,
                                     n_embed=16, proj_ref_path=path, cache_proj_ref=True, **params)


@pytest.fixture(scope="module")
def refs515():
    g = torch.Generator().manual_seed(0)
    return orp.channel_normalise(torch.randn(515, 4, 64, 64, generator=g))


def test_projection_properties_full_size(refs515, tmp_path):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(5, 4, 64, 64, generator=g)
    x[2] = refs515[100] * 1.2                                            # one query near a reference
    base = dict(sigma=3.15, beta_threshold=2.0, beta_threshold_margin=1.6)
    p = make_proc(refs515, tmp_path, scale=0.33, **base)
    a = x.clone().cuda()
    neg, den, isneg = p.conditioning_device(a, want_neg=True)
    # (1) x_out = x - scale * neg exactly as reported; neg is a convex-ish combination: |neg|_pixel <= 1 (refs are unit per pixel)
    torch.testing.assert_close(a.cpu(), x - 0.33 * neg.cpu(), rtol=1e-6, atol=1e-6)
    assert float(torch.linalg.vector_norm(neg, dim=1).max()) <= 1.0 + 1e-4
    # (2) scale = 0 leaves the query untouched but reports the same denominator / gate
    p0 = make_proc(refs515, tmp_path, scale=0.0, **base)
    b = x.clone().cuda()
    _, den0, isneg0 = p0.conditioning_device(b)
    assert torch.equal(b.cpu(), x) and torch.equal(den0, den) and torch.equal(isneg0, isneg)
    # (3) batching: every row equals its single-query run (bitwise: same kernels, same summation order per row)
    for i in (0, 2, 4):
        c = x[i:i + 1].clone().cuda()
        _, d1, _ = p.conditioning_device(c)
        torch.testing.assert_close(c[0], a[i], rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(d1[0], den[i], rtol=2e-6, atol=0)
    # (4) permuting the reference rows changes nothing beyond fp32 summation order
    perm = torch.randperm(515, generator=g)
    pp = make_proc(refs515[perm].contiguous(), tmp_path, scale=0.33, **base)
    d = x.clone().cuda()
    _, den_p, _ = pp.conditioning_device(d)
    torch.testing.assert_close(d, a, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(den_p, den, rtol=1e-5, atol=0)
    # (5) the near-reference query has the largest denominator; denominators are in (eps, M + eps]
    assert int(den.argmax()) == 2 and float(den.min()) > 1e-8 and float(den.max()) <= 515 + 1e-3


def test_maximum_and_empty_reference_sets(tmp_path):
    """M = 3200 is the reference's cap (data/dataloader.py:64-65); M = 0 must be a clean no-op (den = epsilon)."""
    g = torch.Generator().manual_seed(2)
    refs = orp.channel_normalise(torch.randn(3200, 4, 64, 64, generator=g))
    p = make_proc(refs, tmp_path, sigma=3.15, scale=0.33, beta_threshold=1.0)
    x = torch.randn(2, 4, 64, 64, generator=g).cuda()
    x0 = x.clone()
    neg, den, _ = p.conditioning_device(x, want_neg=True)
    assert torch.isfinite(x).all() and torch.isfinite(den).all()
    torch.testing.assert_close(x, x0 - 0.33 * neg, rtol=1e-6, atol=1e-6)
    # empty set through the C ABI directly
    L = sda.lib()
    q = torch.randn(2, 4 * 64 * 64, device="cuda")
    keep = q.clone()
    n = L.sdn_repel_workspace_bytes(2, 0, 4, 4096)
    ws = torch.empty(max(n, 256), dtype=torch.uint8, device="cuda")
    den = torch.empty(2, device="cuda"); flag = torch.empty(2, dtype=torch.int32, device="cuda")
    prm = _lib.RepelParams(n_query=2, n_ref=0, channels=4, hw=4096, weight_fn=0, qnorm=0, sigma=3.15, radius=0.0, scale=0.33,
                           epsilon=1e-8, gate=0.0)
    _lib.check(L.sdn_repel_apply(C.byref(prm), q.data_ptr(), None, None, den.data_ptr(), flag.data_ptr(), ws.data_ptr(),
                                 ws.numel(), _lib.stream_ptr()), "empty refs")
    assert torch.equal(q, keep) and torch.allclose(den.cpu(), torch.full((2,), 1e-8)) and flag.tolist() == [1, 1]


@pytest.mark.parametrize("cls", [DDPMScheduler, DDIMScheduler])
def test_scheduler_round_trip_full_size(cls):
    """step(add_noise(x0, n, t), eps = n).pred_original_sample == x0 for every timestep of the 50-step grid."""
    s = cls(); s.set_timesteps(50)
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(8, 4, 64, 64, generator=g).cuda(); n = torch.randn(8, 4, 64, 64, generator=g).cuda()
    for t in s.timesteps.tolist()[::7] + [1]:
        xt = s.add_noise(x0, n, t)
        out = s.step(n, t, xt, generator=torch.Generator(device="cuda").manual_seed(0))
        torch.testing.assert_close(out.pred_original_sample, x0, rtol=2e-4, atol=2e-4)
    # DDIM is deterministic; its last step lands on sqrt(acp_0) x0 + sqrt(1 - acp_0) eps
    if cls is DDIMScheduler:
        a = s.step(n, 981, x0).prev_sample
        assert torch.equal(a, s.step(n, 981, x0).prev_sample)


@pytest.mark.parametrize("N,M,C,S", [(1, 1, 4, 4), (5, 63, 4, 8), (17, 64, 16, 4), (33, 65, 4, 8), (65, 130, 4, 16), (130, 515, 4, 16),
                                     (16, 7, 4, 64), (64, 40, 16, 8)])
def test_batched_projection_matches_the_oracle_per_query(tmp_path, N, M, C, S):
    """The matrix-core sweeps tile queries by 16 / 32 / 64 (and groups of 64), references by 64 and columns by 64-column
    slices: every tile boundary (ragged N, M, a single reference, several query groups) against the per-query CPU oracle,
    RBF (threshold flavour, sigma from the YAML) and SPARSE."""
    g = torch.Generator().manual_seed(N * 1000 + M)
    refs = orp.channel_normalise(torch.randn(M, C, S, S, generator=g))
    x = torch.randn(N, C, S, S, generator=g) * 0.5
    x[N // 2] = refs[M // 2] * 1.1 + 0.01 * x[N // 2]                       # one query close to a reference
    params = dict(sigma=3.15, scale=0.33, beta_threshold=1.0, beta_threshold_margin=0.25)
    p = make_proc(refs, tmp_path, **params)
    a = x.clone().cuda()
    neg, den, isneg = p.conditioning_device(a, want_neg=True)
    for i in sorted({0, N // 2, N - 1}):
        o = orp.kernel_fast_conditioning(x[i:i + 1].clone(), refs, flavour="threshold", use_beta_threshold=True, **params)
        torch.testing.assert_close(a[i:i + 1].cpu(), o["x_0_hat"], rtol=2e-5, atol=2e-6)
        torch.testing.assert_close(den[i].cpu(), torch.tensor(o["mean_x_0_hat"]["denominator"], dtype=torch.float32), rtol=2e-5, atol=0)
        assert bool(isneg[i].item()) == o["is_negation"]
    radius = float(torch.cdist(x.reshape(N, -1), refs.reshape(M, -1)).median())
    ps = make_proc(refs, tmp_path, method="sparse", radius=radius, scale=0.03)
    b = x.clone().cuda()
    _, _, isn = ps.conditioning_device(b)
    for i in sorted({0, N // 2, N - 1}):
        o = orp.sparse_conditioning(x[i:i + 1].clone(), refs, flavour="threshold", radius=radius, scale=0.03)
        torch.testing.assert_close(b[i:i + 1].cpu(), o["x_0_hat"], rtol=2e-5, atol=2e-5)
        assert bool(isn[i].item()) == o["is_negation"]
