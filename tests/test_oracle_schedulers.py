"""Self-consistency KATs for the scheduler oracle (parity unpinned at the reference level: diffusers is absent)
and oracle <-> product agreement of the host-side coefficient tables."""
import math

import pytest
import torch

from oracle import schedulers as osch
from safe_denoiser_amd import schedulers as psch


def test_ddpm_timesteps_and_tables():
    s = osch.DDPM()
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(981, 0, -20))
    assert abs(float(s.betas[0]) - 0.00085) < 1e-9 and abs(float(s.betas[-1]) - 0.012) < 1e-8
    assert 0.0 < float(s.alphas_cumprod[-1]) < 0.01 and float(s.alphas_cumprod[0]) > 0.999
    # repellency window 780 <= t <= 1000 -> 11 of 50 steps
    assert sum(1 for t in s.timesteps.tolist() if 780 <= t <= 1000) == 11


@pytest.mark.parametrize("cls", [osch.DDPM, osch.DDIM])
def test_x0_recovered_from_exact_noise(cls):
    s = cls()
    s.set_timesteps(50)
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(1, 4, 8, 8, generator=g)
    n = torch.randn(1, 4, 8, 8, generator=g)
    for t in (981, 501, 1):
        xt = s.add_noise(x0, n, t)
        out = s.step(n, t, xt, generator=torch.Generator().manual_seed(1))
        torch.testing.assert_close(out.pred_original_sample, x0, rtol=2e-4, atol=2e-4)


def test_ddim_is_deterministic_and_ddpm_draws_one_randn():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 4, 8, 8, generator=g)
    e = torch.randn(1, 4, 8, 8, generator=g)
    d = osch.DDIM(); d.set_timesteps(50)
    a = d.step(e, 981, x).prev_sample
    b = d.step(e, 981, x).prev_sample
    assert torch.equal(a, b)
    p = osch.DDPM(); p.set_timesteps(50)
    g1 = torch.Generator().manual_seed(7)
    out = p.step(e, 981, x, generator=g1)
    g2 = torch.Generator().manual_seed(7)
    z = torch.randn(e.shape, generator=g2)
    mean = out.prev_sample - p.variance(981) ** 0.5 * z
    out2 = p.step(e, 981, x, generator=torch.Generator().manual_seed(8))
    mean2 = out2.prev_sample - p.variance(981) ** 0.5 * torch.randn(e.shape, generator=torch.Generator().manual_seed(8))
    torch.testing.assert_close(mean, mean2, rtol=1e-5, atol=1e-6)
    # last step (t=1): prev_t < 0 -> variance clamps to 1e-20, noise weight 1e-10, a randn is still drawn
    assert float(p.variance(1)) == pytest.approx(1e-20)
    assert torch.equal(g1.get_state(), g2.get_state())


def test_flow_euler_grid_and_terminal_step():
    f = osch.FlowMatchEuler()
    f.set_timesteps(50)
    assert float(f.sigmas[0]) == pytest.approx(1.0) and float(f.sigmas[-1]) == 0.0
    assert f.sigma_min == pytest.approx(3 * 0.001 / (1 + 2 * 0.001), rel=1e-5)
    inside = [i for i, t in enumerate(f.timesteps.tolist()) if 780 <= t <= 1000]
    assert inside == list(range(len(inside))) and 20 <= len(inside) <= 40
    x = torch.randn(1, 16, 4, 4); v = torch.randn(1, 16, 4, 4)
    f._i = 49                                                  # last step: sigma_next = 0 -> lands on x0 = x - sigma v
    s = f.sigmas[49]
    torch.testing.assert_close(f.step(v, None, x), x - s * v, rtol=1e-5, atol=1e-6)


def test_flow_renoise_formula():
    g = torch.Generator().manual_seed(0)
    x, v, z = (torch.randn(1, 16, 4, 4, generator=g) for _ in range(3))
    s, sn = 0.9, 0.85
    out = osch.flow_repellency_renoise(x, v, s, sn, lambda a: a * 0.5, z)
    x0r = (x - s * v) * 0.5
    noise = math.sqrt(sn) * (x + (1 - s) * v) + math.sqrt(1 - sn) * z
    torch.testing.assert_close(out, x0r + sn * (noise - x0r), rtol=1e-6, atol=1e-6)


def test_product_tables_match_oracle():
    o, p = osch.DDPM(), psch.DDPMScheduler()
    o.set_timesteps(50); p.set_timesteps(50)
    assert torch.equal(o.alphas_cumprod, p.alphas_cumprod) and o.timesteps.tolist() == p.timesteps.tolist()
    for t in o.timesteps.tolist():
        co = p.step_coefficients(t)
        assert co["sigma"] == pytest.approx(float(o.variance(t) ** 0.5), rel=1e-6)
        a_t = o.alphas_cumprod[t]
        assert co["sqrt_ac"] == pytest.approx(float(a_t ** 0.5), rel=1e-7)
    di, pi = osch.DDIM(), psch.DDIMScheduler()
    di.set_timesteps(50); pi.set_timesteps(50)
    co = pi.step_coefficients(1)
    assert co["c_x0"] == pytest.approx(float(di.alphas_cumprod[0] ** 0.5)) and co["sigma"] == 0.0
    fo, fp = osch.FlowMatchEuler(), psch.FlowMatchEulerDiscreteScheduler()
    fo.set_timesteps(50); fp.set_timesteps(50)
    assert torch.equal(fo.sigmas, fp.sigmas) and torch.equal(fo.timesteps, fp.timesteps)
