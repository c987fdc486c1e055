"""HIP scheduler / guidance kernels vs the CPU oracle, same seeds.  fp32 elementwise: rtol 1e-6, atol 1e-6."""
import math

import pytest
import torch

import safe_denoiser_amd as sda
from oracle import schedulers as osch
from safe_denoiser_amd import _lib, schedulers as psch

pytestmark = pytest.mark.gpu


def close(a, b, rt=2e-6, at=2e-6):
    torch.testing.assert_close(a.cpu(), b.cpu(), rtol=rt, atol=at)


@pytest.mark.parametrize("shape", [(1, 4, 64, 64), (3, 4, 8, 8), (515, 4, 64, 64)])
def test_ddpm_step_add_noise_x0(shape):
    g = torch.Generator().manual_seed(0)
    x, e, z = (torch.randn(shape, generator=g) for _ in range(3))
    o, p = osch.DDPM(), psch.DDPMScheduler()
    o.set_timesteps(50); p.set_timesteps(50, device="cuda")
    for t in (981, 781, 21, 1):
        class G:                                                   # feed the oracle the same z
            pass
        exp = o.step(e, t, x, generator=None) if False else None
        co = p.step_coefficients(t)
        out = p.step(e.cuda(), t, x.cuda(), noise=z.cuda())
        a_t = o.alphas_cumprod[t]
        x0 = (x - (1 - a_t) ** 0.5 * e) / a_t ** 0.5
        close(out.pred_original_sample, x0, rt=1e-5, at=1e-5)
        mean = co["c_x0"] * x0 + co["c_x"] * x
        close(out.prev_sample, mean + float(o.variance(t) ** 0.5) * z, rt=1e-5, at=1e-5)
        close(p.add_noise(x.cuda(), z.cuda(), t), o.add_noise(x, z, t))


def test_ddpm_generator_draw_order_matches_torch():
    """step() must consume exactly one randn(model_output.shape) from the device generator per call (t > 0)."""
    p = psch.DDPMScheduler(); p.set_timesteps(50, device="cuda")
    x = torch.randn(1, 4, 64, 64, device="cuda"); e = torch.randn(1, 4, 64, 64, device="cuda")
    g1 = torch.Generator(device="cuda").manual_seed(11)
    a = p.step(e, 981, x, generator=g1).prev_sample
    g2 = torch.Generator(device="cuda").manual_seed(11)
    z = torch.randn(e.shape, generator=g2, device="cuda", dtype=torch.float32)
    b = p.step(e, 981, x, noise=z).prev_sample
    assert torch.equal(a, b)
    assert torch.equal(g1.get_state(), g2.get_state())


def test_ddim_step():
    g = torch.Generator().manual_seed(1)
    x, e = (torch.randn(2, 4, 64, 64, generator=g) for _ in range(2))
    o, p = osch.DDIM(), psch.DDIMScheduler()
    o.set_timesteps(50); p.set_timesteps(50, device="cuda")
    for t in (981, 501, 1):
        exp = o.step(e, t, x)
        out = p.step(e.cuda(), t, x.cuda())
        close(out.prev_sample, exp.prev_sample, rt=1e-5, at=1e-5)
        close(out.pred_original_sample, exp.pred_original_sample, rt=1e-5, at=1e-5)


@pytest.mark.parametrize("nb", [2, 3])
def test_cfg_combine(nb):
    g = torch.Generator().manual_seed(2)
    P, D = 5, 4 * 64 * 64
    mo = torch.randn(nb * P, D, generator=g)
    u, t = mo[:P], mo[P:2 * P]
    exp = u + 7.5 * (t - u)
    mog = mo.cuda(); out = torch.empty(P, D, device="cuda")
    _lib.check(sda.lib().sdn_cfg_combine(mog.data_ptr(), P, nb, D, 7.5, out.data_ptr(), _lib.stream_ptr()), "cfg")
    close(out, exp)


def test_renoise_select_is_per_prompt():
    g = torch.Generator().manual_seed(3)
    P, D = 6, 4 * 64 * 64
    lat, x0r, z = (torch.randn(P, D, generator=g) for _ in range(3))
    flags = torch.tensor([1, 0, 0, 1, 1, 0], dtype=torch.int32)
    sa, s1 = 0.3, 0.95
    exp = torch.where(flags.bool()[:, None], sa * x0r + s1 * z, lat)
    lg, xg, zg, fg = lat.clone().cuda(), x0r.cuda(), z.cuda(), flags.cuda()     # keep the device tensors alive
    _lib.check(sda.lib().sdn_renoise_select(lg.data_ptr(), xg.data_ptr(), zg.data_ptr(), fg.data_ptr(), P, D, sa, s1,
                                            _lib.stream_ptr()), "renoise")
    close(lg, exp)


def test_flow_kernels():
    g = torch.Generator().manual_seed(4)
    x, v, z = (torch.randn(2, 16, 64, 64, generator=g) for _ in range(3))
    o, p = osch.FlowMatchEuler(), psch.FlowMatchEulerDiscreteScheduler()
    o.set_timesteps(50); p.set_timesteps(50, device="cuda")
    for _ in range(3):
        close(p.step(v.cuda(), None, x.cuda()).prev_sample, o.step(v, None, x), rt=1e-5, at=1e-5)
    # fp16 in -> fp32 math -> fp16 out
    outh = p.step(v.half().cuda(), None, x.half().cuda()).prev_sample
    assert outh.dtype == torch.float16
    s, sn = float(o.sigmas[5]), float(o.sigmas[6])
    exp = osch.flow_repellency_renoise(x, v, s, sn, lambda a: a * 0.5, z)
    xg, vg = x.cuda(), v.cuda()
    x0 = torch.empty_like(xg); x1 = torch.empty_like(xg); out = torch.empty_like(xg)
    L = sda.lib()
    _lib.check(L.sdn_flow_endpoints(xg.data_ptr(), vg.data_ptr(), xg.numel(), s, x0.data_ptr(), x1.data_ptr(),
                                    _lib.stream_ptr()), "endpoints")
    close(x0, x - s * v); close(x1, x + (1 - s) * v)
    x0r = (x0 * 0.5).contiguous()
    zg = z.cuda()
    _lib.check(L.sdn_flow_renoise(x0r.data_ptr(), x1.data_ptr(), zg.data_ptr(), xg.numel(), sn, out.data_ptr(),
                                  _lib.stream_ptr()), "renoise")
    close(out, exp, rt=1e-5, at=1e-5)


def test_sld_guidance_kernel_with_momentum_state():
    """SLD eq. 3-8 over three steps (momentum carried), warm-up boundary included; fp32 elementwise."""
    g = torch.Generator().manual_seed(6)
    P, D = 3, 4 * 64 * 64
    mom = torch.zeros(P, D); mg = mom.clone().cuda()
    cfg = dict(scale=2000.0, thr=0.025, ms=0.5, mb=0.7)
    for step in range(3):
        mo = torch.randn(3 * P, D, generator=g) * 0.01 + torch.randn(1, D, generator=g)
        u, t, c = mo[:P], mo[P:2 * P], mo[2 * P:]
        scale = torch.clamp((t - c).abs() * cfg["scale"], max=1.0)
        scale = torch.where((t - c) >= cfg["thr"], torch.zeros_like(scale), scale)
        gs = (c - u) * scale + cfg["ms"] * mom
        mom = cfg["mb"] * mom + (1 - cfg["mb"]) * gs
        apply = step >= 1
        exp = u + 7.5 * ((t - u) - (gs if apply else 0))
        mog = mo.cuda(); out = torch.empty(P, D, device="cuda")
        _lib.check(sda.lib().sdn_sld_guidance(mog.data_ptr(), P, D, 7.5, cfg["scale"], cfg["thr"], cfg["ms"], cfg["mb"],
                                              int(apply), mg.data_ptr(), out.data_ptr(), _lib.stream_ptr()), "sld")
        close(out, exp, rt=1e-5, at=1e-5)
        close(mg, mom, rt=1e-5, at=1e-6)


@pytest.mark.parametrize("nb", [2, 3])
def test_cfg_combine_rows_is_the_scalar_kernel_per_prompt(nb):
    """One guidance scale per prompt (run_nudity.py:390-396 reads it row by row): bit-identical to the scalar kernel run on
    each prompt with its own scale, and equal to the oracle formula."""
    g = torch.Generator().manual_seed(12)
    P, D = 5, 4 * 64 * 64
    mo = torch.randn(nb * P, D, generator=g)
    gs = torch.tensor([7.5, 9.0, 3.0, 7.5, 12.25])
    exp = mo[:P] + gs[:, None] * (mo[P:2 * P] - mo[:P])
    mog, gg = mo.cuda(), gs.cuda()
    out = torch.empty(P, D, device="cuda")
    L = sda.lib()
    _lib.check(L.sdn_cfg_combine_rows(mog.data_ptr(), P, nb, D, gg.data_ptr(), out.data_ptr(), _lib.stream_ptr()), "cfg rows")
    close(out, exp)
    for p in range(P):
        one = torch.cat([mog[p:p + 1], mog[P + p:P + p + 1]] + ([mog[2 * P + p:2 * P + p + 1]] if nb == 3 else [])).contiguous()
        ref = torch.empty(1, D, device="cuda")
        _lib.check(L.sdn_cfg_combine(one.data_ptr(), 1, nb, D, float(gs[p]), ref.data_ptr(), _lib.stream_ptr()), "cfg")
        assert torch.equal(ref[0], out[p])
    assert L.sdn_cfg_combine_rows(mog.data_ptr(), P, nb, D, None, out.data_ptr(), _lib.stream_ptr()) != 0   # null scales refused
    assert L.sdn_cfg_combine_rows(mog.data_ptr(), 0, nb, D, gg.data_ptr(), out.data_ptr(), _lib.stream_ptr()) == 0


def test_sld_guidance_rows_is_the_scalar_kernel_per_prompt():
    g = torch.Generator().manual_seed(13)
    P, D = 4, 4 * 32 * 32
    gs = torch.tensor([7.5, 5.0, 9.0, 7.5])
    cfg = (1000.0, 0.01, 0.3, 0.4)
    L = sda.lib()
    mom_rows = torch.zeros(P, D, device="cuda")
    mom_one = [torch.zeros(1, D, device="cuda") for _ in range(P)]
    gg = gs.cuda()
    for step in range(3):
        mo = (torch.randn(3 * P, D, generator=g) * 0.01 + torch.randn(1, D, generator=g)).cuda()
        out = torch.empty(P, D, device="cuda")
        _lib.check(L.sdn_sld_guidance_rows(mo.data_ptr(), P, D, gg.data_ptr(), *cfg, int(step >= 1), mom_rows.data_ptr(),
                                           out.data_ptr(), _lib.stream_ptr()), "sld rows")
        for p in range(P):
            one = torch.cat([mo[p:p + 1], mo[P + p:P + p + 1], mo[2 * P + p:2 * P + p + 1]]).contiguous()
            ref = torch.empty(1, D, device="cuda")
            _lib.check(L.sdn_sld_guidance(one.data_ptr(), 1, D, float(gs[p]), *cfg, int(step >= 1), mom_one[p].data_ptr(),
                                          ref.data_ptr(), _lib.stream_ptr()), "sld")
            assert torch.equal(ref[0], out[p]) and torch.equal(mom_one[p][0], mom_rows[p])
