"""The batched HIP loop vs the per-prompt CPU oracle loop on the SAME per-prompt noise tapes (seed-for-seed).

Tolerance: final latents, relative L2 per prompt vs the oracle run with bf16 storage emulation <= LOOP_BOUND[mode] = the
measured distance + 25 % (2.2e-2 ... 3.9e-2 measured; rounds 1-2 used a blanket 8e-2) (measured
3.5e-2 .. 4.1e-2 after 20 steps; the tests run 12).  The loop arithmetic is fp32 on both sides; the residue is the UNet's bf16 storage
noise (1.1e-2 per forward, see test_gpu_unet.py) amplified by classifier-free guidance (eps = u + 7.5 (t - u)
multiplies uncorrelated errors of the two branches by ~10) and integrated over the trajectory.  The is_negation
decisions / number of re-noise draws must match exactly (a mismatch would shift the random stream)."""
import pytest
import torch

from oracle import pipeline as opipe
from oracle import repellency as orp
from oracle import schedulers as osch
from oracle.unet import OracleUNet
from safe_denoiser_amd.pipeline import SafeDenoiserPipeline
from safe_denoiser_amd.repellency import repellency_methods_fast as fast
from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
from safe_denoiser_amd.schedulers import DDIMScheduler, DDPMScheduler
from safe_denoiser_amd.unet import UNet2DConditionModel

pytestmark = pytest.mark.gpu

SMALL = dict(block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
             layers_per_block=1, attention_head_dim=8, cross_attention_dim=768, sample_size=16)
SMALL_O = dict(block_out_channels=(320, 640), level_has_attn=(True, False), layers_per_block=1, n_heads=8,
               cross_dim=768, sample_size=16)
STEPS = 10            # DDPM / DDIM leading spacing: t = 901, 801 fall in the 780..1000 window (the CPU oracle sets the test time)


class Tapes:
    """Per-prompt pre-generated noise, served in draw order; independent cursors for oracle and engine."""

    def __init__(self, n_prompts, shape, n_draws, seed):
        g = torch.Generator().manual_seed(seed)
        self.data = [torch.randn((n_draws,) + tuple(shape), generator=g) for _ in range(n_prompts)]
        self.cur = [0] * n_prompts

    def __call__(self, p, shape):
        z = self.data[p][self.cur[p]].reshape(shape)
        self.cur[p] += 1
        return z.clone()


# Where the oracle's torch ops are evaluated: the CPU.  Round 5 tried the GPU for these loops (on a shared box the small CPU loops
# took 30-45 s each under host contention) and found that a STORAGE-EMULATING oracle is not the same function on the two devices:
# fp32 summation order differs in the last bit, which moves 16-bit roundings, and over a 10-step CFG loop the two evaluations of
# the SAME bf16-emulating oracle end 3.4e-2 apart -- as far as the engine is from either (test_oracle_on_both_devices below pins
# the pure-fp32 oracle across devices at <= 1e-4 and records the emulating one's spread).  So the 16-bit loop bounds of this file are
# distances between two bf16-noisy trajectories (a statistic, measured + 25 %), the full-size tests against the pure-fp32 oracle
# (tests/test_gpu_f32.py, test_gpu_e2e_ids.py) are the acceptance, and the emulating oracles stay on the device the bounds were
# measured on.  ODEV = "cuda" would evaluate them on the GPU.
ODEV = "cpu"


class DevTapes:
    """A tape served on a device, sharing the cursor with the host tape it wraps."""

    def __init__(self, tapes, dev):
        self.t, self.dev = tapes, dev

    @property
    def cur(self):
        return self.t.cur

    def __call__(self, p, shape):
        return self.t(p, shape).to(self.dev)


def o_unet(sd, act_dtype, dev=None):
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    return OracleUNet(sd, SMALL_O, act_dtype=act_dtype, device=dev or ODEV)


def o_repel(repel, dev=None):
    if repel is None:
        return None
    return {k: (v.to(dev or ODEV) if torch.is_tensor(v) else v) for k, v in repel.items()}


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def world():
    u = UNet2DConditionModel(text_len=77, **SMALL)
    sd = u.synthetic_state_dict(11)
    u.load_state_dict(sd)
    g = torch.Generator().manual_seed(2)
    P = 2
    E = torch.randn(2 * P, 77, 768, generator=g)
    refs = orp.channel_normalise(torch.randn(24, 4, 16, 16, generator=g))
    return u, sd, E, refs, P


def make_proc(mod, refs, tmp_path, **params):
    path = str(tmp_path / f"pr_{abs(hash(str(params))) % 10**9}.pt")
    torch.save(refs, path)
    return mod.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085, 0.012,
                                     n_embed=4, proj_ref_path=path, cache_proj_ref=True, **params)


def run_oracle(sd, E, refs, P, tapes, sched, variant, repel, dev=None):
    dev = dev or ODEV
    unet = o_unet(sd, torch.bfloat16, dev)
    outs, draws = [], 0
    for p in range(P):
        pair = torch.stack([E[p], E[P + p]]).to(dev)
        lat, st = opipe.denoise_one(unet, sched(), pair, p, DevTapes(tapes, dev), num_inference_steps=STEPS, repel=o_repel(repel, dev),
                                    variant=variant)
        outs.append(lat.cpu())
        draws += st["renoise_draws"]
    return torch.cat(outs), draws


# Loop bounds = the distance measured on MI355X (round 3, gpurun_out/t_all.log: max over the prompts of each mode) + 25 %
# (VERDICT r2 #1d; the blanket 8e-2 / 1.5e-2 of rounds 1-2 left a 2-4x slack).  16-bit storage vs the storage-emulating
# oracle at the small configuration; CFG 7.5 amplifies the two branches' uncorrelated rounding ~3x over one forward.
LOOP_BOUND = {"ddpm_threshold_time": 4.6e-2, "ddim_threshold_time": 4.5e-2, "ddpm_time_fast": 4.8e-2, "ddpm_norepel": 3.5e-2,
              "ddim_norepel": 2.8e-2, "ddpm_sparse": 4.6e-2, "fp16": 4.2e-3, "sld": 5.0e-2, "lra_svf": 3.1e-2, "lra_re_attn": 4.5e-2,
              "svf_2branch": 3.2e-2, "safree_call": 3.1e-2}


@pytest.mark.parametrize("mode", ["ddpm_threshold_time", "ddim_threshold_time", "ddpm_time_fast", "ddpm_norepel",
                                  "ddim_norepel",          # BASELINE config 1: DDIM 50-step plumbing with repellency OFF
                                  "ddpm_sparse"])          # row R4 (SPELL) inside the loop
def test_loop_matches_oracle(world, tmp_path, mode):
    u, sd, E, refs, P = world
    shape = (1, 4, 16, 16)
    sched_o, sched_p = (osch.DDIM, DDIMScheduler) if mode.startswith("ddim") else (osch.DDPM, DDPMScheduler)
    if mode.endswith("threshold_time"):
        # gate chosen between the denominators the prompts actually produce, so both branches are taken
        probe = Tapes(P, shape, 3 * STEPS + 4, seed=5)
        dens = []
        unet = OracleUNet(sd, SMALL_O, act_dtype=torch.bfloat16)
        for p in range(P):
            s = sched_o(); s.set_timesteps(STEPS)
            lat = probe(p, shape)
            out = unet(torch.cat([lat] * 2), 901.0, torch.stack([E[p], E[P + p]]))
            eps = out[0:1] + 7.5 * (out[1:2] - out[0:1])
            x0 = s.step(eps, 901, lat, generator=torch.Generator().manual_seed(0)).pred_original_sample
            _, den, _ = orp.kernel_fast_score(x0, refs, 3.15, 1e-8)
            dens.append(float(den))
        srt = sorted(dens)
        gate = 0.5 * (srt[-1] + srt[-2]) if srt[-1] - srt[-2] > 1e-3 * srt[-1] else srt[-1] * 0.9
        params = dict(sigma=3.15, scale=0.33, beta_threshold=gate + 1.6, beta_threshold_margin=1.6)
        repel_o = dict(flavour="threshold", proj_refs=refs, **params)
        proc, variant = make_proc(thr, refs, tmp_path, **params), "threshold_time"
    elif mode == "ddpm_time_fast":
        params = dict(scale=0.33)
        repel_o = dict(flavour="fast", proj_refs=refs, **params)
        proc, variant = make_proc(fast, refs, tmp_path, **params), "time"
    elif mode == "ddpm_sparse":
        # radius between the distances the x0 probes actually have to the references, so some prompts find neighbours
        # (is_negation, re-noise) and others do not; scale of configs/sparse_repellency/spell.yaml
        probe = Tapes(P, shape, 3 * STEPS + 4, seed=5)
        unet = OracleUNet(sd, SMALL_O, act_dtype=torch.bfloat16)
        dmin = []
        for p in range(P):
            s_ = sched_o(); s_.set_timesteps(STEPS)
            lat = probe(p, shape)
            out = unet(torch.cat([lat] * 2), 901.0, torch.stack([E[p], E[P + p]]))
            eps = out[0:1] + 7.5 * (out[1:2] - out[0:1])
            x0 = s_.step(eps, 901, lat, generator=torch.Generator().manual_seed(0)).pred_original_sample
            dmin.append(float(torch.cdist(x0.reshape(1, -1), refs.reshape(len(refs), -1)).min()))
        srt = sorted(dmin)
        radius = 0.5 * (srt[0] + srt[1]) if srt[1] - srt[0] > 1e-3 * srt[1] else srt[0] * 1.05
        params = dict(radius=radius, scale=0.03)
        repel_o = dict(flavour="threshold", method="sparse", proj_refs=refs, **params)
        path = str(tmp_path / "pr_sparse.pt")
        torch.save(refs, path)
        proc = thr.get_repellency_method("sparse", torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085, 0.012,
                                         n_embed=4, proj_ref_path=path, cache_proj_ref=True, **params)
        variant = "threshold_time"
    else:
        repel_o, proc, variant = None, None, "threshold_time"

    t_o = Tapes(P, shape, 3 * STEPS + 4, seed=5)
    ref, draws_o = run_oracle(sd, E, refs, P, t_o, sched_o, variant, repel_o)
    t_p = Tapes(P, shape, 3 * STEPS + 4, seed=5)
    pipe = SafeDenoiserPipeline(u, sched_p(), variant=variant)
    lat = pipe(prompt_embeddings=E.cuda(), num_inference_steps=STEPS, guidance_scale=7.5, repellency_processor=proc,
               noise_fn=t_p, return_latents=True)
    torch.cuda.synchronize()
    assert torch.isfinite(lat).all()
    assert t_p.cur == t_o.cur, (t_p.cur, t_o.cur)                      # identical draw order per prompt
    assert pipe.last_stats["renoise_draws"] == draws_o
    errs = [rel_l2(lat[p:p + 1], ref[p:p + 1]) for p in range(P)]
    print(f"{mode}: renoise draws {draws_o}, per-prompt rel L2 {['%.2e' % e for e in errs]}")
    assert max(errs) <= LOOP_BOUND[mode], errs
    if mode.endswith("threshold_time") or mode == "ddpm_sparse":
        assert draws_o > 0                                             # the gate fired at least once


def test_oracle_on_both_devices(world):
    """The oracle is a CPU restatement in plain torch ops; the full-size tests evaluate those ops on the GPU (TF32 off).  Pin, on the
    whole loop of one prompt (5 steps, one of them in the repellency window) (UNet, CFG, x0 probe, repellency, re-noise, DDPM steps): the PURE-fp32 oracle lands on the same
    latents on either device (<= 1e-4).  Recorded beside it: the bf16-EMULATING oracle does not -- last-bit differences of the fp32
    sums move its 16-bit roundings and the loop amplifies them to the size of the bf16 noise itself (3.4e-2 measured over 10 steps, profiles/round5_gpu_suite_oracle_devices.txt) -- which is
    why the emulating oracles of this file stay on the CPU, where their bounds were measured."""
    u, sd, E, refs, P = world
    shape = (1, 4, 16, 16)
    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)
    res = {}
    for name, act in (("fp32", None), ("bf16-emulating", torch.bfloat16)):
        out = {}
        for dev in ("cpu", "cuda"):
            t = Tapes(P, shape, 3 * STEPS + 4, seed=5)
            out[dev], st = opipe.denoise_one(o_unet(sd, act, dev), osch.DDPM(), torch.stack([E[0], E[P]]).to(dev), 0, DevTapes(t, dev),
                                             num_inference_steps=5, repel=o_repel(dict(flavour="threshold", proj_refs=refs, **params), dev))
            assert st["renoise_draws"] == 1
        res[name] = rel_l2(out["cuda"], out["cpu"])
    print(f"oracle loop, GPU evaluation vs CPU evaluation (small configuration): pure fp32 {res['fp32']:.2e}, bf16-emulating {res['bf16-emulating']:.2e}")
    assert res["fp32"] <= 1e-4
    assert res["bf16-emulating"] <= 8e-2          # a sanity bound only: see the docstring


def test_device_generators_are_per_prompt(world):
    """With real generators: prompt p's result does not depend on which other prompts share the batch."""
    u, sd, E, refs, P = world
    gens = lambda idx: [torch.Generator(device="cuda").manual_seed(1000 + i) for i in idx]
    pipe = SafeDenoiserPipeline(u, DDPMScheduler(), variant="threshold_time")
    full = pipe(prompt_embeddings=E.cuda(), num_inference_steps=5, generator=gens(range(P)), return_latents=True)
    sel = [P - 1]
    Esel = torch.cat([E[:P][sel], E[P:][sel]])
    one = pipe(prompt_embeddings=Esel.cuda(), num_inference_steps=5, generator=gens(sel), return_latents=True)
    assert rel_l2(one, full[P - 1:P]) <= 1e-6
    with pytest.raises(NotImplementedError):
        pipe(prompt="a photo", num_inference_steps=5, return_latents=True)


def test_loop_with_engine_side_latent_repeat_is_bit_identical(world, tmp_path):
    """A UNet built with latent_repeat = 2 takes the P latents directly (no cat([latents] * 2) copy) and shares the
    branch-independent prefix: the loop's latents must not change by a single bit."""
    u, sd, E, refs, P = world
    u2 = UNet2DConditionModel(text_len=77, latent_repeat=2, **SMALL)
    u2.load_state_dict(sd)
    outs = []
    for unet in (u, u2):
        proc = make_proc(thr, refs, tmp_path, scale=0.03, sigma=1.0, epsilon=1e-8, beta_threshold=0.5, beta_threshold_margin=0.1)
        pipe = SafeDenoiserPipeline(unet, DDPMScheduler(), variant="threshold_time")
        gens = [torch.Generator(device="cuda").manual_seed(50 + i) for i in range(P)]
        outs.append(pipe(prompt_embeddings=E.cuda(), num_inference_steps=8, generator=gens, repellency_processor=proc, return_latents=True))
    torch.testing.assert_close(outs[1], outs[0], rtol=0, atol=0)
    with pytest.raises(Exception):                                   # 3 branches on a repeat-2 plan
        SafeDenoiserPipeline(u2, DDPMScheduler())(prompt_embeddings=torch.cat([E, E[:P]]).cuda(), num_inference_steps=2,
                                                   sld_guidance_scale=2000.0, return_latents=True)


def test_fp16_storage_loop_parity(tmp_path):
    """Same tape test with fp16 storage: final latents within 1.5e-2 rel L2 (bf16: 2.5-4e-2)."""
    u = UNet2DConditionModel(text_len=77, dtype=torch.float16, **SMALL)
    sd = u.synthetic_state_dict(11)
    u.load_state_dict(sd)
    g = torch.Generator().manual_seed(2)
    P = 2
    E = torch.randn(2 * P, 77, 768, generator=g)
    shape = (1, 4, 16, 16)
    unet_o = o_unet(sd, torch.float16)
    t_o = Tapes(P, shape, 3 * STEPS + 4, seed=9)
    ref = torch.cat([opipe.denoise_one(unet_o, osch.DDPM(), torch.stack([E[p], E[P + p]]).to(ODEV), p, DevTapes(t_o, ODEV),
                                       num_inference_steps=STEPS)[0].cpu() for p in range(P)])
    t_p = Tapes(P, shape, 3 * STEPS + 4, seed=9)
    lat = SafeDenoiserPipeline(u, DDPMScheduler())(prompt_embeddings=E.cuda(), num_inference_steps=STEPS, noise_fn=t_p, return_latents=True)
    errs = [rel_l2(lat[p:p + 1], ref[p:p + 1]) for p in range(P)]
    print(f"fp16 loop: per-prompt rel L2 {['%.2e' % e for e in errs]}")
    assert t_p.cur == t_o.cur and max(errs) <= LOOP_BOUND["fp16"]


def test_sld_family_loop_matches_oracle(world, tmp_path):
    """SLD pipelines (modified_sld_pipeline_threshold_time.py): 3 branches [uncond | text | safety concept], guidance
    eq. 3-8 with momentum, then the same repellency window.  fp32 guidance math; bf16 UNet -> 8e-2 bound as above."""
    u, sd, E, refs, P = world
    g = torch.Generator().manual_seed(9)
    concept = torch.randn(P, 77, 768, generator=g)
    E3 = torch.cat([E, concept])                                   # [3P,77,768]
    sld = dict(scale=1000.0, warmup=2, thr=0.02, ms=0.3, mb=0.4)
    shape = (1, 4, 16, 16)
    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)
    t_o = Tapes(P, shape, 3 * STEPS + 4, seed=13)
    unet = o_unet(sd, torch.bfloat16)
    ref = torch.cat([opipe.denoise_one(unet, osch.DDPM(), torch.stack([E3[p], E3[P + p], E3[2 * P + p]]).to(ODEV), p, DevTapes(t_o, ODEV),
                                       num_inference_steps=STEPS, repel=o_repel(dict(flavour="threshold", proj_refs=refs, **params)),
                                       sld=sld)[0].cpu() for p in range(P)])
    t_p = Tapes(P, shape, 3 * STEPS + 4, seed=13)
    pipe = SafeDenoiserPipeline(u, DDPMScheduler(), variant="threshold_time")
    lat = pipe(prompt_embeddings=E3.cuda(), num_inference_steps=STEPS, repellency_processor=make_proc(thr, refs, tmp_path, **params),
               noise_fn=t_p, sld_guidance_scale=sld["scale"], sld_warmup_steps=sld["warmup"], sld_threshold=sld["thr"],
               sld_momentum_scale=sld["ms"], sld_mom_beta=sld["mb"], return_latents=True)
    errs = [rel_l2(lat[p:p + 1], ref[p:p + 1]) for p in range(P)]
    print(f"sld loop: per-prompt rel L2 {['%.2e' % e for e in errs]}")
    assert t_p.cur == t_o.cur and max(errs) <= LOOP_BOUND["sld"]


# ---- SAFREE text switching inside the loop, 3-branch `lra` batches (...threshold_time.py:518-548) -------------------
def _safe_text(E, P, seed):
    """A stand-in for the SAFREE-projected embeddings: same unconditional rows, perturbed text rows."""
    g = torch.Generator().manual_seed(seed)
    Es = E.clone()
    Es[P:] = E[P:] + 1.5 * torch.randn(P, 77, 768, generator=g)
    return Es


@pytest.mark.parametrize("mode", ["lra_svf", "lra_re_attn", "svf_2branch"])
def test_lra_and_safree_text_switch_match_oracle(world, tmp_path, mode):
    """`lra` = three branches [uncond | E' | text_e] per prompt (the third is computed and discarded, as the reference
    does); svf: prompt p uses the projected text while i <= beta_adjusted[p] (a DIFFERENT step count per prompt, so the
    batch is mixed); otherwise the re_attn_t step window.  Same tapes, same draw counts, bf16 bound as above."""
    u, sd, E, refs, P = world
    shape = (1, 4, 16, 16)
    Es = _safe_text(E, P, 77)
    lra = mode.startswith("lra")
    if mode.endswith("svf") or mode == "svf_2branch":
        betas = [3, 0, 7][:P]                                            # different step counts: the batch is mixed
        sf = dict(safree=True, svf=True, lra=lra, re_attn_t=(-1, -1))
        fn = lambda p: (lambda i: i <= betas[p])
    else:
        betas = None
        sf = dict(safree=True, svf=False, lra=lra, re_attn_t=(2, 6))
        fn = lambda p: (lambda i: 2 <= i <= 6)
    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)        # gate always fires
    unet = o_unet(sd, torch.bfloat16)
    rep_o = o_repel(dict(flavour="threshold", proj_refs=refs, **params))
    t_o = Tapes(P, shape, 3 * STEPS + 4, seed=21)
    outs, draws = [], 0
    for p in range(P):
        lat, st = opipe.denoise_one(unet, osch.DDPM(), torch.stack([E[p], E[P + p]]).to(ODEV), p, DevTapes(t_o, ODEV), num_inference_steps=STEPS,
                                    repel=rep_o, lra=lra, text_safe=torch.stack([Es[p], Es[P + p]]).to(ODEV), use_safe_fn=fn(p))
        outs.append(lat.cpu()); draws += st["renoise_draws"]
    ref = torch.cat(outs)
    # control: the switch matters (the oracle WITHOUT it lands elsewhere)
    t_c = Tapes(P, shape, 3 * STEPS + 4, seed=21)
    ctl = opipe.denoise_one(unet, osch.DDPM(), torch.stack([E[0], E[P]]).to(ODEV), 0, DevTapes(t_c, ODEV), num_inference_steps=STEPS,
                            repel=rep_o, lra=lra)[0].cpu()
    t_p = Tapes(P, shape, 3 * STEPS + 4, seed=21)
    pipe = SafeDenoiserPipeline(u, DDPMScheduler(), variant="threshold_time")
    lat = pipe(prompt_embeddings=E.cuda(), num_inference_steps=STEPS, repellency_processor=make_proc(thr, refs, tmp_path, **params),
               noise_fn=t_p, safree_dict=sf, rescaled_text_embeddings=Es.cuda(), beta_adjusted=betas, return_latents=True)
    assert pipe.last_stats["branches"] == (3 if lra else 2)
    assert t_p.cur == t_o.cur and pipe.last_stats["renoise_draws"] == draws > 0
    errs = [rel_l2(lat[p:p + 1], ref[p:p + 1]) for p in range(P)]
    sep = rel_l2(ctl, ref[0:1])
    print(f"{mode}: per-prompt rel L2 {['%.2e' % e for e in errs]}; without the text switch prompt 0 is {sep:.2e} away")
    assert max(errs) <= LOOP_BOUND[mode] and sep > 1.5 * max(errs)


def test_window_kwargs_follow_each_variant(world, tmp_path):
    """`*_threshold.py` hard-codes its step-index window and never reads negation_warmup_start/end (...threshold.py:430-431);
    modified_stable_diffusion_pipeline_threshold_time.py reads them as step-INDEX bounds (start = lower);
    `*_threshold_time.py` as timestep bounds (start = upper)."""
    u, sd, E, refs, P = world
    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)
    n = 10
    def windows(variant, **kw):
        pipe = SafeDenoiserPipeline(u, DDIMScheduler(), variant=variant)
        pipe(prompt_embeddings=E.cuda(), num_inference_steps=n, repellency_processor=make_proc(thr, refs, tmp_path, **params), **kw, return_latents=True)
        return pipe.last_stats["window_steps"]
    assert windows("threshold", negation_warmup_start=1000, negation_warmup_end=780) == n       # kwargs ignored: every step
    assert windows("sd_threshold_time") == n                                                   # i = 0..9 all <= 11
    assert windows("sd_threshold_time", negation_warmup_start=2, negation_warmup_end=4) == 3   # i in {2, 3, 4}
    assert windows("threshold_time") == 2                                                      # t = 901, 801 are >= 780
    assert windows("threshold_time", negation_warmup_start=700, negation_warmup_end=400) == 3  # t = 601, 501, 401


def test_eliding_the_dead_lra_branch_gives_the_same_bits(world, tmp_path):
    """`lra`: the reference computes a third guidance branch and discards its output (...threshold_time.py:542-544).  With
    elide_dead_branch=True the engine runs the two live branches only: identical latents, identical draw counts."""
    u, sd, E, refs, P = world
    Es = _safe_text(E, P, 77)
    sf = dict(safree=True, svf=True, lra=True, re_attn_t=(-1, -1))
    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)
    outs, draws = [], []
    for elide in (False, True):
        pipe = SafeDenoiserPipeline(u, DDPMScheduler(), variant="threshold_time", elide_dead_branch=elide)
        outs.append(pipe(prompt_embeddings=E.cuda(), num_inference_steps=STEPS, repellency_processor=make_proc(thr, refs, tmp_path, **params),
                         noise_fn=Tapes(P, (1, 4, 16, 16), 3 * STEPS + 4, seed=21), safree_dict=sf, rescaled_text_embeddings=Es.cuda(),
                         beta_adjusted=[3, 0, 7][:P], return_latents=True))
        draws.append(pipe.last_stats["renoise_draws"])
        assert pipe.last_stats["branches"] == (2 if elide else 3)
    assert draws[0] == draws[1] > 0
    torch.testing.assert_close(outs[1], outs[0], rtol=0, atol=0)


def test_mixed_guidance_batch_gives_each_prompt_the_bits_of_its_own_scalar_call(world, tmp_path):
    """a prompt table with a `guidance` column (run_nudity.py:390-396) no longer fragments batches --
    `guidance_scale=[g_0 .. g_{P-1}]` runs sdn_cfg_combine_rows.  Same tapes -> prompt p's latents are bit-identical to a
    one-prompt call with scalar guidance g_p (the default plans give a sample the same bits at every batch size)."""
    u, sd, E, refs, P = world
    shape = (1, 4, 16, 16)
    proc = make_proc(thr, refs, tmp_path, scale=0.33, sigma=3.15, beta_threshold=1e-6, beta_threshold_margin=1e9)
    pipe = SafeDenoiserPipeline(u, DDPMScheduler(), variant="threshold_time")
    gs = [9.0, 5.5]
    tapes = Tapes(P, shape, 3 * STEPS + 4, seed=21)
    both = pipe(prompt_embeddings=E, num_inference_steps=STEPS, guidance_scale=gs, repellency_processor=proc, noise_fn=tapes,
                return_latents=True)
    assert pipe.last_stats["renoise_draws"] > 0
    for p in range(P):
        t1 = Tapes(P, shape, 3 * STEPS + 4, seed=21)
        one = pipe(prompt_embeddings=torch.stack([E[p], E[P + p]]), num_inference_steps=STEPS, guidance_scale=gs[p],
                   repellency_processor=proc, noise_fn=lambda _p, sh, p=p, t1=t1: t1(p, sh), return_latents=True)
        assert torch.equal(one[0], both[p]), p
    with pytest.raises(Exception):
        pipe(prompt_embeddings=E, num_inference_steps=2, guidance_scale=[7.5], return_latents=True)


def test_precision_schedule_switches_plans_per_step(world, tmp_path):
    """Round 5: `SafeDenoiserPipeline(unet=<16-bit plan>, unet_hi=<precise plan>, precision_schedule=...)` runs each step on the plan
    the schedule names.  Plumbing check at the small configuration with the readme's three-branch text switching (per-prompt svf step
    counts -> the mixed text buffer exists on BOTH plans) and a firing gate:
      * "all"  == the precise plan alone, bit for bit;  "none" == the 16-bit plan alone, bit for bit;
      * {"window": True} = the repellency-window steps only: its latents are NOT the all-16-bit ones, and sit closer to the precise
        plan's than the 16-bit plan's do (at this size the window is 2 of 10 steps -- the ones that carry the error);
      * draw counts and the tapes' cursors never depend on the schedule."""
    u_bf, sd, E, refs, P = world
    shape = (1, 4, 16, 16)
    Es = _safe_text(E, P, 77)
    sf = dict(safree=True, svf=True, lra=True, re_attn_t=(-1, -1))
    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)
    lo = UNet2DConditionModel(text_len=77, dtype=torch.float16, latent_repeat=3, **SMALL); lo.load_state_dict(sd)
    hi = UNet2DConditionModel(text_len=77, precision="bf16x3", latent_repeat=3, **SMALL); hi.load_state_dict(sd)

    def run(unet, unet_hi=None, schedule=None):
        pipe = SafeDenoiserPipeline(unet, DDPMScheduler(), variant="threshold_time", unet_hi=unet_hi, precision_schedule=schedule)
        t = Tapes(P, shape, 3 * STEPS + 4, seed=21)
        lat = pipe(prompt_embeddings=E.cuda(), num_inference_steps=STEPS, repellency_processor=make_proc(thr, refs, tmp_path, **params),
                   noise_fn=t, safree_dict=sf, rescaled_text_embeddings=Es.cuda(), beta_adjusted=[3, 0, 7][:P], return_latents=True)
        return lat, pipe.last_stats, list(t.cur)

    only_lo, st_lo, cur_lo = run(lo)
    only_hi, st_hi, cur_hi = run(hi)
    s_all, st_all, cur_all = run(lo, hi, "all")
    s_none, st_none, cur_none = run(lo, hi, "none")
    s_win, st_win, cur_win = run(lo, hi, {"window": True})
    assert torch.equal(s_all, only_hi) and torch.equal(s_none, only_lo)
    assert st_all["hi_steps"] == STEPS and st_none["hi_steps"] == 0 and st_win["hi_steps"] == st_win["window_steps"] == 2
    assert cur_lo == cur_hi == cur_all == cur_none == cur_win
    assert st_lo["renoise_draws"] == st_hi["renoise_draws"] == st_win["renoise_draws"] > 0
    d_lo, d_win = rel_l2(only_lo, only_hi), rel_l2(s_win, only_hi)
    print(f"small configuration, distance from the bf16x3 plan's latents: fp16 plan {d_lo:.2e}, fp16 + bf16x3 inside the window {d_win:.2e}")
    assert not torch.equal(s_win, only_lo) and d_win < 0.6 * d_lo
    # a later call on the 16-bit plan alone is unaffected by the second plan's text cache (versions are process-wide)
    again, _, _ = run(lo)
    assert torch.equal(again, only_lo)


def test_a_batch_a_few_prompts_over_whole_waves_runs_split_and_gives_the_same_latents(world, tmp_path):
    """65 prompts x 3 branches (what three of the eight ranks of the 515-prompt job run): `pipe.tail_split` (default on) sends the
    64 aligned prompts and the 65th through the UNet as two concurrent forwards per step (unet._tail_split_of).  Same latents, bit
    for bit, as the one-forward-per-step loop -- with the readme's three-branch text switching (per-prompt svf step counts: plain,
    projected and MIXED text buffers all pass through the split path's regather), a firing gate and the sync-free window."""
    _, sd, _, refs, _ = world
    P, steps = 65, 6
    g = torch.Generator().manual_seed(65)
    E = torch.randn(2 * P, 77, 768, generator=g)
    Es = _safe_text(E, P, 78)
    sf = dict(safree=True, svf=True, lra=True, re_attn_t=(-1, -1))
    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)
    u = UNet2DConditionModel(text_len=77, latent_repeat=3, **SMALL); u.load_state_dict(sd)
    beta = [(3 * p) % 5 for p in range(P)]                                          # svf step counts 0 .. 4: mixed batches at steps 1 .. 4

    def run(split):
        pipe = SafeDenoiserPipeline(u, DDPMScheduler(), variant="threshold_time")
        pipe.tail_split = split
        gens = [torch.Generator(device="cuda").manual_seed(1000 + p) for p in range(P)]
        lat = pipe(prompt_embeddings=E.cuda(), num_inference_steps=steps, repellency_processor=make_proc(thr, refs, tmp_path, **params),
                   generator=gens, safree_dict=sf, rescaled_text_embeddings=Es.cuda(), beta_adjusted=beta, return_latents=True)
        return lat, pipe.last_stats, [int(g_.get_offset()) for g_ in gens]

    a, st_a, off_a = run(False)
    b, st_b, off_b = run(True)
    assert st_a["tail_split"] is None and st_b["tail_split"] == (64, 1)
    assert st_a["renoise_draws"] == st_b["renoise_draws"] > 0 and off_a == off_b
    assert torch.equal(a, b)
    c, _, _ = run(False)                                                             # and back: the plan's own text cache is not confused
    assert torch.equal(a, c)


def test_discarded_lra_branch_runs_on_the_16_bit_plan_at_precise_steps(world, tmp_path):
    """`pipe.dead_branch_lo` (default on): at the steps a precision schedule sends to `unet_hi`, the two live guidance branches run on the
    precise plan and `lra`'s discarded third branch on the 16-bit plan (the reference computes it and drops it, ...threshold_time.py:
    542-544: its precision cannot reach the latents).  Same latents as running all three branches on the precise plan -- to the last
    fp32 bits of a differently shaped launch (<= 1e-5; the bound a split forward of the bf16x3 plan gets everywhere) --, same draw
    counts and tape cursors; without `lra` (two branches, both live) nothing changes."""
    _, sd, E, refs, P = world
    shape = (1, 4, 16, 16)
    Es = _safe_text(E, P, 77)
    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)
    lo = UNet2DConditionModel(text_len=77, dtype=torch.float16, latent_repeat=3, **SMALL); lo.load_state_dict(sd)
    hi = UNet2DConditionModel(text_len=77, precision="bf16x3", latent_repeat=3, **SMALL); hi.load_state_dict(sd)

    def run(split, u_lo=lo, u_hi=hi, lra=True):
        sf = dict(safree=True, svf=True, lra=lra, re_attn_t=(-1, -1))
        pipe = SafeDenoiserPipeline(u_lo, DDPMScheduler(), variant="threshold_time", unet_hi=u_hi, precision_schedule={"window": True})
        pipe.dead_branch_lo = split
        t = Tapes(P, shape, 3 * STEPS + 4, seed=21)
        lat = pipe(prompt_embeddings=E.cuda(), num_inference_steps=STEPS, repellency_processor=make_proc(thr, refs, tmp_path, **params),
                   noise_fn=t, safree_dict=sf, rescaled_text_embeddings=Es.cuda(), beta_adjusted=[3, 0, 7][:P], return_latents=True)
        return lat, pipe.last_stats, list(t.cur)

    a, st_a, cur_a = run(False)
    b, st_b, cur_b = run(True)
    assert st_a["dead_branch_lo_steps"] == 0 and st_b["dead_branch_lo_steps"] == st_b["hi_steps"] == st_b["window_steps"] == 2
    assert cur_a == cur_b and st_a["renoise_draws"] == st_b["renoise_draws"] > 0
    d = rel_l2(b, a)
    print(f"discarded branch on the 16-bit plan at the precise steps: latents vs all three on the precise plan {d:.2e}")
    assert d <= 1e-5
    lo2 = UNet2DConditionModel(text_len=77, dtype=torch.float16, latent_repeat=2, **SMALL); lo2._weights = lo._weights
    hi2 = UNet2DConditionModel(text_len=77, precision="bf16x3", latent_repeat=2, **SMALL); hi2._weights = hi._weights
    c, st_c, _ = run(True, lo2, hi2, lra=False)
    e, st_e, _ = run(False, lo2, hi2, lra=False)
    assert st_c["dead_branch_lo_steps"] == 0 and st_c["branches"] == 2 and torch.equal(c, e)
