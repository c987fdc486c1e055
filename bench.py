#!/usr/bin/env python3
"""Headline benchmark: images/sec (512x512, 50 steps, SD-v1.4 + repellency) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one batch of P prompts taken END TO END through the README-default configuration of BASELINE config 2
(configs/base/vanilla/safree_neg_prompt_config.json:26-28: safree = svf = lra = true; README.md:216-226), exactly as
`run_nudity.py:439-460` calls the pipeline:
    token ids -> CLIP text encoder (prompts, negative prompt, the 17 concept phrases, one masked copy per prompt token)
    -> SAFREE projection + self-validation filter (per-prompt beta_adjusted) -> 50 denoising iterations, each = UNet forward
    on [3P,4,64,64] ([uncond | projected-or-plain text | text], per-prompt text switching) + guidance combine + (t in
    780..1000: x0 probe + repellency projection against proj_ref[515] + device-side re-noise select) + DDPM step
    -> VAE decode -> uint8 [P,512,512,3] images (on the device; PIL wrapping is host-side file I/O).
`value` = images/sec of that whole call.  Inputs are synthetic (no weights / datasets / vocabulary files on the box) and
resident in HBM before the timed region: SD-v1.4-architecture UNet / CLIP / VAE with random weights generated on the GPU,
synthetic prompt strings through a hash tokenizer (the real tokenizer is the caller's vocabulary lookup), proj_ref =
channel-normalised randn([515,4,64,64], seed 0), repellency knobs of configs/nudity/safe_denoiser.yaml, beta_threshold
calibrated by the engine's own row-R5 path.  Prompts shard across ranks (no collective in the loop); rank 0 broadcasts
proj_ref + threshold once over RCCL.  Prints ONE JSON line on rank 0.  `--workload latent` times the round-1/2 headline
instead (prompt embeddings in, latents out, 2 CFG branches); at N = 1 that figure is also reported as a secondary.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0          # HBM3E peak, same table


NEG_SPACE = ["Sexual Acts", "Content Meant to Arouse Sexual Excitement", "Nudity", "Pornography", "Erotic Art", "Lustful",
             "Seductive", "Orgasmic", "Libido", "Kinky", "Sexual Orientation", "Sexual Attraction", "Sexual Intercourse",
             "Sexual Pleasure", "Sexual Fantasy", "Carnal Desires", "Sexual Gratification"]        # run_nudity.py:353-358
SAFREE = dict(safree=True, svf=True, lra=True, alpha=0.01, up_t=10, re_attn_t=[-1, 1001], category="nudity", logger=None)


def synthetic_prompt(i: int) -> str:
    """Prompt i of the synthetic 515-prompt job: 6..18 pseudo-words (i2p prompts are of that length), deterministic."""
    n = 6 + (i * 7) % 13
    return " ".join(f"w{(i * 131 + j * 17) % 997}" for j in range(n))


def _dtype(args):
    return torch.float16 if args.dtype == "f16" else torch.bfloat16


def _enc_kw(args):
    return dict(dtype=_dtype(args)) if args.text_precision == "same" else dict(precision=args.text_precision)


def build_engine(args, rank, world, dev):
    from safe_denoiser_amd import dist as sdist
    from safe_denoiser_amd.pipeline import SafeDenoiserPipeline, make_scheduler
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
    from safe_denoiser_amd.unet import UNet2DConditionModel

    e2e = args.workload == "e2e"
    # latent_repeat = number of guidance branches: the engine-side form of the reference's cat([latents] * n) -- the branches
    # share their latents, so the UNet computes the branch-independent prefix once (bit-identical, tests/test_gpu_unet.py)
    nb = 3 if e2e else 2
    unet = UNet2DConditionModel(dtype=_dtype(args), latent_repeat=1 if args.no_latent_repeat else nb)
    unet.load_synthetic_on_device(1234, device=dev)
    sched = make_scheduler(args.scheduler)
    vae = enc = tok = None
    if e2e:
        from safe_denoiser_amd.clip import CLIPTextModel
        from safe_denoiser_amd.vae import AutoencoderKL
        from tests_support.fake_tokenizer import FakeCLIPTokenizer
        # the text encoder runs in bf16x3 in the HEADLINE too (round 5): its output feeds the categorical SAFREE decisions -- a bf16
        # encoder flips one for 1 prompt in 8 against the fp32 chain -- and it is ~0.5 % of a batch (--text-precision same: 16-bit)
        enc = CLIPTextModel(**_enc_kw(args))
        enc.load_synthetic_on_device(4242, device=dev)
        vae = AutoencoderKL(dtype=_dtype(args))
        vae.load_synthetic_on_device(4321, device=dev)
        tok = FakeCLIPTokenizer()

    refs = None
    if rank == 0:
        g = torch.Generator().manual_seed(0)
        refs = torch.randn(args.refs, 4, 64, 64, generator=g)
        refs = refs / torch.norm(refs, dim=1, keepdim=True)
    comm_ms = sdist.warm_up_communicator(dev)                          # RCCL's lazy communicator set-up, timed on its own
    sdist.barrier(); torch.cuda.synchronize()
    tb0 = time.perf_counter()
    refs = sdist.broadcast_proj_ref(refs, dev)                         # RCCL broadcast, 33.75 MB (+ checksum all-reduces)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - tb0) * 1e3 if world > 1 else None    # N = 1: there is no broadcast (a host -> device copy)
    sdist.heartbeat(f"communicator {comm_ms:.0f} ms, proj_ref broadcast {bcast_ms if bcast_ms is None else round(bcast_ms, 1)} ms")
    tmp = tempfile.mkdtemp(prefix=f"sdn_bench_r{rank}_")
    path = os.path.join(tmp, "repellency_proj_ref.pt")
    torch.save(refs.cpu(), path)
    knobs = dict(scale=0.33, sigma=3.15, beta_threshold_margin=1.6, proj_ref_path=path, cache_proj_ref=True)
    if rank == 0:                                                       # row R5: calibrate once, share the scalar
        cal = thr.get_repellency_method("kernel_fast", torch.zeros(1, device=dev), None, None, 50, 1000, 0.00085,
                                        0.012, n_embed=16, scheduler=make_scheduler("ddpm"),
                                        proj_noisy_ref_path_for_beta=None, **knobs)
        beta = float(cal.beta_threshold)
        del cal
    else:
        beta = 0.0
    beta = sdist.broadcast_scalar(beta, dev)
    proc = thr.get_repellency_method("kernel_fast", torch.zeros(1, device=dev), None, None, 50, 1000, 0.00085, 0.012,
                                     n_embed=16, beta_threshold=beta, **knobs)
    pipe = SafeDenoiserPipeline(unet, sched, variant="threshold_time", vae=vae, text_encoder=enc, tokenizer=tok)
    return unet, pipe, proc, beta, bcast_ms, comm_ms


def _attn_pad(label: str) -> float:
    """issued / algorithmic MFMA work of k_attn<d>: (ceil16(d) + ceil32(d + (d % 32 != 0))) / (2 d)."""
    d = int(label[label.index("<") + 1:label.index(">")])
    qk = -(-d // 16) * 16
    pv = -(-(d + (1 if d % 32 else 0)) // 32) * 32
    return (qk + pv) / (2.0 * d)


def label_symbol(label: str, tname: str):
    """Plan label -> (display name of the kernel SYMBOL, regex matching its demangled name in the rocprof / PMC records).
    Several labels share one symbol (`k_gemm<10>` and `k_gemm<10>/rp` -- residual into the accumulators -- are both
    k_gemm_dma<T, 10, 4, 2, 0>); `/lnL` carries the symbol's LNF parameter, `k_conv_slab<W>` its map width.  WGM follows from NREP
    (>= 8: the 256-row tile); NSTAGE is 2 except for small grids of the NREP = 5 tile, which the label does not tell apart."""
    import re
    t = f"Sdn{tname}"
    m = re.fullmatch(r"k_gemm<(\d+)>(/rp|/ln(\d)|/s\d+|x3)?", label)
    if m and m.group(2) != "x3":
        nrep, lnf = int(m.group(1)), int(m.group(3) or 0)
        wgm = 4 if nrep >= 8 else 2
        if (m.group(2) or "").startswith("/s"):
            return label, None                                          # split-K forms (small batches only)
        st = r"\d+" if nrep == 5 else "2"
        return f"k_gemm_dma<{t}, {nrep}, {wgm}, {'2|4' if nrep == 5 else '2'}, {lnf}>", rf"k_gemm_dma<{t}, {nrep}, {wgm}, {st}, {lnf}>"
    m = re.fullmatch(r"k_conv_slab<(\d+)>", label)
    if m:
        return f"k_conv_slab<{t}, {m.group(1)}>", rf"k_conv_slab<{t}, {m.group(1)}>"
    m = re.fullmatch(r"k_attn<(\d+)>", label)
    if m:                                       # one label, two instantiations (query sets per wave: self- / cross-attention)
        return f"k_attn<{t}, {m.group(1)}, ...>", rf"k_attn<{t}, {m.group(1)},"
    if label == "k_ffn320":
        return f"k_ffn320<{t}>", rf"k_ffn320<{t}>"
    return label, None


def _pmc_records(pattern: str, lib_sha: str):
    """Newest profiles/round*_<pattern>.json whose libsdn sha equals the running library's -> (record | None, source note)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"round*_{pattern}.json")),
                   key=lambda f: int("".join(ch for ch in os.path.basename(f).split("_")[0] if ch.isdigit()) or 0))
    if not cands:
        return None, f"no profiles/round*_{pattern}.json record"
    rec = json.load(open(cands[-1]))
    meta = rec.get("__meta__", {})
    if meta.get("libsdn_sha256") != lib_sha:
        return None, (f"profiles/{os.path.basename(cands[-1])} was collected on another build of libsdn.so (record "
                      f"{str(meta.get('libsdn_sha256'))[:12]}, running {lib_sha[:12]}): stale, not reported")
    return rec, (f"profiles/{os.path.basename(cands[-1])} (git {(meta.get('git_head') or '?')[:10]}, same libsdn.so as this run, "
                 f"batch {meta.get('batch', '?')}; {meta.get('collected_with', '')})")


def cpu_baseline(args):
    """The CPU oracle (a port of the reference loop, fp32 torch ops) on the host cores, on a BOUNDED sample of the timed
    workload: TWO of the 50 denoising iterations of ONE prompt at the full SD-v1.4 size -- t = 981 inside the repellency
    window (UNet on the 3 guidance branches + CFG + x0 probe + repellency M + re-noise + DDPM step) and t = 761 outside it
    (UNet + CFG + DDPM step) -- plus one VAE decode of that prompt's latent, timed separately and extrapolated to the loop's
    11 window + 39 plain iterations + 1 decode.  (The CLIP / SAFREE front end is < 0.1 % of the work and is left out.)"""
    from oracle import repellency as orp
    from oracle import schedulers as osch
    from oracle.unet import OracleUNet
    from oracle.vae import OracleVAEDecoder
    from safe_denoiser_amd.unet import UNet2DConditionModel
    from safe_denoiser_amd.vae import AutoencoderKL
    cores = torch.get_num_threads()
    nb = 3 if args.workload == "e2e" else 2
    sd = UNet2DConditionModel().synthetic_state_dict(1234)
    unet = OracleUNet(sd, None, act_dtype=None)
    g = torch.Generator().manual_seed(0)
    refs = orp.channel_normalise(torch.randn(args.refs, 4, 64, 64, generator=g))
    lat = torch.randn(1, 4, 64, 64, generator=g)
    text = torch.randn(nb, 77, 768, generator=g)
    s = osch.DDPM(); s.set_timesteps(50)
    t0 = time.perf_counter()
    unet(lat, 500.0, text[:1])                                         # untimed: pages the 3.4 GB of weights in, spins the thread pool up
    t_warm = time.perf_counter() - t0
    times = {}
    for t in (981, 761):
        t0 = time.perf_counter()
        out = unet(torch.cat([lat] * nb), float(t), text)
        eps = out[0:1] + 7.5 * (out[1:2] - out[0:1])
        if t >= 780:
            x0 = s.step(eps, t, lat, generator=g).pred_original_sample
            d = orp.kernel_fast_conditioning(x0, refs, flavour="threshold", scale=0.33, sigma=3.15, beta_threshold=1e-6,
                                             beta_threshold_margin=1e9, use_beta_threshold=True)
            lat = s.add_noise(d["x_0_hat"], torch.randn(lat.shape, generator=g), t)
        lat = s.step(eps, t, lat, generator=g).prev_sample
        times[t] = time.perf_counter() - t0
    # SURVEY 8d's own CPU case, BASELINE config 1 (SD-v1.4 DDIM 50 steps, 2 CFG branches, repellency OFF, fp32): one iteration of one
    # prompt at full size, extrapolated to 50 (on the resident oracle UNet)
    cfg1 = None
    try:
        unet1 = unet
        sd_ = osch.DDIM(); sd_.set_timesteps(50)
        lat1 = torch.randn(1, 4, 64, 64, generator=g)
        t0 = time.perf_counter()
        out1 = unet1(torch.cat([lat1] * 2), 981.0, text[:2])
        eps1 = out1[0:1] + 7.5 * (out1[1:2] - out1[0:1])
        lat1 = sd_.step(eps1, 981, lat1).prev_sample
        t_it = time.perf_counter() - t0
        cfg1 = {"value": 1.0 / (50 * t_it), "unit": "images/sec", "sample": f"BASELINE config 1 (DDIM 50 steps, b = 2, repellency off, fp32): one full-size "
                                                                            f"iteration of one prompt = {t_it:.2f} s, extrapolated x 50"}
    except Exception as e:                                            # the secondary figure must never cost the line
        cfg1 = {"error": repr(e)}
    t_dec = 0.0
    if args.workload == "e2e":
        del unet, sd
        dec = OracleVAEDecoder(AutoencoderKL().synthetic_state_dict(4321), None, act_dtype=None)
        t0 = time.perf_counter()
        dec.decode_latents(lat * 0.18215)
        t_dec = time.perf_counter() - t0
    per_image = 11 * times[981] + 39 * times[761] + t_dec
    return {"value": 1.0 / per_image, "unit": "images/sec", "cores": cores, "kind": "port",
            "threads": {"torch_get_num_threads": cores, "torch_get_num_interop_threads": torch.get_num_interop_threads(),
                        "os_cpu_count": os.cpu_count(), "sched_affinity": len(os.sched_getaffinity(0)),
                        "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS")},
            "warmup_forward_s": t_warm, "config1": cfg1,
            "sample": f"1 prompt x 2 of 50 iterations at full size: t=981 {times[981]:.2f} s (window: UNet b={nb} fp32 + CFG + x0 "
                      f"probe + repellency M={args.refs} + re-noise + DDPM step), t=761 {times[761]:.2f} s (UNet + CFG + DDPM step)"
                      + (f", VAE decode {t_dec:.2f} s" if t_dec else "") +
                      f" = {times[981] + times[761] + t_dec:.1f} s of CPU work; extrapolated 11 x window + 39 x plain"
                      + (" + 1 decode" if t_dec else "")}


def measure_latent(args, dev, proc, P, precision=None, steps_timed=1, inference_steps=None):
    """Secondary figures on the round-1/2 headline workload (prompt embeddings in, final latents out, 2 CFG branches,
    80.3 TFLOP per image) with the same gate: the 16-bit engine (precision None) or a precision mode ("bf16x3" / "fp32":
    fp32 storage; the modes that meet the north star's 1e-3).  One warm-up call of 3 iterations + `steps_timed` timed
    batches of P prompts."""
    from safe_denoiser_amd.pipeline import SafeDenoiserPipeline, make_scheduler
    from safe_denoiser_amd.unet import UNet2DConditionModel
    kw = dict(precision=precision) if precision else dict(dtype=_dtype(args))
    u = UNet2DConditionModel(latent_repeat=2, **kw)
    u.load_synthetic_on_device(1234, device=dev)
    pipe = SafeDenoiserPipeline(u, make_scheduler(args.scheduler), variant="threshold_time")
    n = inference_steps or args.inference_steps
    g = torch.Generator().manual_seed(7)
    text = torch.randn(P, 77, 768, generator=g)
    uncond = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(8))
    E = torch.cat([uncond.expand(P, -1, -1), text]).to(dev)
    gens = lambda: [torch.Generator(device=dev).manual_seed(1000 + i) for i in range(P)]
    pipe(prompt_embeddings=E, num_inference_steps=3, guidance_scale=7.5, generator=gens(), repellency_processor=proc,
         return_latents=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps_timed):
        out = pipe(prompt_embeddings=E, num_inference_steps=n, guidance_scale=7.5, generator=gens(), repellency_processor=proc,
                   return_latents=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps_timed
    assert torch.isfinite(out).all()
    fl, _ = u.flops(2 * P)
    res = {"value": P / dt, "unit": "images/sec", "workload": "prompt embeddings -> latents, 2 CFG branches (the round-1/2 headline)",
           "dtype": precision or args.dtype, "prompts_per_batch": P, "timed_batches": steps_timed, "inference_steps": n,
           "renoise_draws": pipe.last_stats["renoise_draws"], "unet_tflops_algorithmic": fl * n / dt / 1e12}
    del u, pipe
    torch.cuda.empty_cache()
    return res


def measure_e2e_elided(args, dev, proc, P, mine):
    """The headline call with `elide_dead_branch=True`: `lra`'s third guidance branch, whose output the reference discards
    (...threshold_time.py:542-544), is not computed -- bit-identical images (tests/test_gpu_pipeline.py), 2/3 of the UNet work.
    NOT the headline: `value` keeps the reference's three branches.  One warm-up (3 iterations) + one timed batch."""
    from safe_denoiser_amd.clip import CLIPTextModel
    from safe_denoiser_amd.pipeline import SafeDenoiserPipeline, make_scheduler
    from safe_denoiser_amd.unet import UNet2DConditionModel
    from safe_denoiser_amd.vae import AutoencoderKL
    from tests_support.fake_tokenizer import FakeCLIPTokenizer
    u = UNet2DConditionModel(dtype=_dtype(args), latent_repeat=2); u.load_synthetic_on_device(1234, device=dev)
    enc = CLIPTextModel(**_enc_kw(args)); enc.load_synthetic_on_device(4242, device=dev)
    vae = AutoencoderKL(dtype=_dtype(args)); vae.load_synthetic_on_device(4321, device=dev)
    pipe = SafeDenoiserPipeline(u, make_scheduler(args.scheduler), variant="threshold_time", vae=vae, text_encoder=enc,
                                tokenizer=FakeCLIPTokenizer(), elide_dead_branch=True)
    idx = [mine[j % len(mine)] for j in range(P)]
    call = lambda n: pipe([synthetic_prompt(i) for i in idx], guidance_scale=7.5, num_inference_steps=n,
                          negative_prompt=", ".join(NEG_SPACE), negative_prompt_space=NEG_SPACE,
                          generator=[torch.Generator(device=dev).manual_seed(1000 + i) for i in idx], repellency_processor=proc,
                          safree_dict=dict(SAFREE), output_type="uint8")
    call(3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = call(args.inference_steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res = {"value": P / dt, "unit": "images/sec", "guidance_branches_computed": pipe.last_stats["branches"], "prompts_per_batch": P,
           "note": "the headline call with lra's dead third branch not computed: bit-identical images, 80.3 TFLOP of UNet work per "
                   "image instead of 120.5; a labelled secondary, not `value`"}
    assert out.dtype == torch.uint8
    del u, enc, vae, pipe
    torch.cuda.empty_cache()
    return res


def measure_e2e_precision(args, dev, proc, P, mine, precision="bf16x3", timed_batches=2):
    """The HEADLINE call itself -- token ids -> CLIP -> SAFREE -> 50 DDPM steps x UNet on three guidance branches + repellency +
    re-noise -> VAE decode -> uint8 images, same gate, same prompts -- in the precision mode that meets the north star's latents
    tolerance FROM TOKEN IDS: UNet and text encoder with fp32 storage and bf16x3 contractions (tests/test_gpu_e2e_ids.py: every
    SAFREE decision agrees with the pure-fp32 chain, final latents 5.4e-5 from it; the 16-bit text encoder flips 1 decision in 8).
    The VAE decoder sits after the parity tap and stays 16-bit.  One warm-up call (3 iterations) + `timed_batches` timed batches."""
    from safe_denoiser_amd.clip import CLIPTextModel
    from safe_denoiser_amd.pipeline import SafeDenoiserPipeline, make_scheduler
    from safe_denoiser_amd.unet import UNet2DConditionModel
    from safe_denoiser_amd.vae import AutoencoderKL
    from tests_support.fake_tokenizer import FakeCLIPTokenizer
    u = UNet2DConditionModel(latent_repeat=3, precision=precision); u.load_synthetic_on_device(1234, device=dev)
    enc = CLIPTextModel(precision=precision); enc.load_synthetic_on_device(4242, device=dev)
    vae = AutoencoderKL(dtype=_dtype(args)); vae.load_synthetic_on_device(4321, device=dev)
    pipe = SafeDenoiserPipeline(u, make_scheduler(args.scheduler), variant="threshold_time", vae=vae, text_encoder=enc,
                                tokenizer=FakeCLIPTokenizer())

    def call(k, n):
        idx = [mine[(k * P + j) % len(mine)] for j in range(P)]
        return pipe([synthetic_prompt(i) for i in idx], num_images_per_prompt=1, guidance_scale=7.5, num_inference_steps=n,
                    negative_prompt=", ".join(NEG_SPACE), negative_prompt_space=NEG_SPACE, height=512, width=512,
                    generator=[torch.Generator(device=dev).manual_seed(1000 + i) for i in idx], repellency_processor=proc,
                    safree_dict=dict(SAFREE), output_type="uint8")
    call(0, 3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    renoise = 0
    for k in range(timed_batches):
        out = call(1 + k, args.inference_steps)
        renoise += pipe.last_stats["renoise_draws"]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / timed_batches
    assert out.dtype == torch.uint8 and tuple(out.shape) == (P, 512, 512, 3)
    fl, _ = u.flops(3 * P)
    res = {"value": P / dt, "unit": "images/sec", "dtype": precision, "workload": "the headline call (README-default end to end, 3 guidance branches)",
           "prompts_per_batch": P, "timed_batches": timed_batches, "ms_per_batch": dt * 1e3, "renoise_draws": renoise,
           "guidance_branches_computed": pipe.last_stats["branches"],
           "precision": "UNet + CLIP text encoder: fp32 storage, bf16x3 split-operand contractions (GEMM operands as bf16 hi|lo|hi "
                        "triples on the LDS-DMA tiles); schedulers / guidance / repellency fp32; VAE decoder (after the parity tap) 16-bit",
           "ids_to_latents_rel_l2_vs_fp32_chain": "5.4e-5 max over 8 prompts, all SAFREE decisions equal (profiles/round4_e2e_ids.json)",
           "unet_tflops_algorithmic_upper_bound": fl * args.inference_steps / dt / 1e12}
    del u, enc, vae, pipe
    torch.cuda.empty_cache()
    return res


def measure_job(args, dev, P, gate_beta, total=None):
    """The job a USER runs: `driver.run_job` -- the body of the reference's main(), run_nudity.py:341-529 -- over
    the whole `--total-prompts`-row prompt table on this one GPU: table read, batching, the pipeline call per batch, and per image
    the PNG writes into {safe|unsafe,all}/ + a (stub) classifier + logs.txt, then detect_dict.json / config.yaml; real files in a
    temporary directory.  The host I/O of batch k overlaps the GPU's batch k + 1 (driver._OrderedWriter).  Everything is built
    from scratch here so that START-UP is timed the way a fresh process pays it: weights (synthetic, generated on the GPU: stands
    where the checkpoint read + upload would be) -> sdn_unet_prepare (derived weight regions) -> first forward at the batch size
    (launch plan + arena + code objects) -> R5 calibration of the gate -> communicator (none at N = 1)."""
    import contextlib
    import shutil
    import yaml
    from safe_denoiser_amd import driver
    from safe_denoiser_amd.clip import CLIPTextModel
    from safe_denoiser_amd.pipeline import SafeDenoiserPipeline, make_scheduler
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
    from safe_denoiser_amd.unet import UNet2DConditionModel
    from safe_denoiser_amd.vae import AutoencoderKL
    from tests_support.fake_tokenizer import FakeCLIPTokenizer
    total = total or args.total_prompts
    sync = torch.cuda.synchronize
    st = {}
    sync(); t0 = time.perf_counter()
    u = UNet2DConditionModel(dtype=_dtype(args), latent_repeat=3); u.load_synthetic_on_device(1234, device=dev)
    enc = CLIPTextModel(**_enc_kw(args)); enc.load_synthetic_on_device(4242, device=dev)
    vae = AutoencoderKL(dtype=_dtype(args)); vae.load_synthetic_on_device(4321, device=dev)
    sync(); t1 = time.perf_counter()
    u._prepare(); enc._prepare(); vae._prepare()                       # (already part of the loads above: re-run to time it on its own)
    sync(); t2 = time.perf_counter()
    st["weights_s"], st["sdn_unet_prepare_s"] = (t1 - t0) - (t2 - t1), t2 - t1
    x = torch.randn(P, 4, 64, 64, device=dev)
    tb = u.prepare_text(torch.randn(3 * P, 77, 768, device=dev))
    y = torch.empty((3 * P, 4, 64, 64), device=dev)
    u.forward_into(x, 981.0, tb, y)
    sync(); t3 = time.perf_counter()
    st["plan_arena_first_forward_s"] = t3 - t2
    del x, tb, y
    tmp = tempfile.mkdtemp(prefix="sdn_job_")
    g = torch.Generator().manual_seed(0)
    refs = torch.randn(args.refs, 4, 64, 64, generator=g)
    refs = refs / torch.norm(refs, dim=1, keepdim=True)
    rpath = os.path.join(tmp, "repellency_proj_ref.pt")
    torch.save(refs, rpath)
    knobs = dict(scale=0.33, sigma=3.15, beta_threshold_margin=1.6, proj_ref_path=rpath, cache_proj_ref=True)
    cal = thr.get_repellency_method("kernel_fast", torch.zeros(1, device=dev), None, None, 50, 1000, 0.00085, 0.012, n_embed=16,
                                    scheduler=make_scheduler("ddpm"), proj_noisy_ref_path_for_beta=None, **knobs)
    r5 = float(cal.beta_threshold)
    sync(); t4 = time.perf_counter()
    st["r5_calibration_s"] = t4 - t3
    st["communicator_s"] = 0.0
    st["startup_s"] = t4 - t0
    del cal
    proc = thr.get_repellency_method("kernel_fast", torch.zeros(1, device=dev), None, None, 50, 1000, 0.00085, 0.012, n_embed=16,
                                     beta_threshold=gate_beta, **knobs)          # the headline's gate (every pair fires on synthetic weights)
    pipe = SafeDenoiserPipeline(u, make_scheduler(args.scheduler), variant="threshold_time", vae=vae, text_encoder=enc,
                                tokenizer=FakeCLIPTokenizer())
    # the prompt table (i2p dialect: case_number, prompt, categories, evaluation_seed) + the reference's three configuration layers
    with open(os.path.join(tmp, "prompts.csv"), "w") as f:
        f.write("case_number,prompt,categories,evaluation_seed\n")
        for i in range(total):
            f.write(f'{i},"{synthetic_prompt(i)}",sexual,{1000 + i}\n')
    task = {"repellency": {"method": "kernel_fast", "n_embed": 16, "guidance_scale": 0.0, "params": {k_: v for k_, v in knobs.items()}},
            "data": {"name": "nudity"}, "mean_processor": {}}
    yaml.safe_dump(task, open(os.path.join(tmp, "task.yaml"), "w"))
    cfg = {"erase_id": "safree_neg_prompt_rep_threshold_time", "nudity": "nudity", "data": os.path.join(tmp, "prompts.csv"),
           "save_dir": os.path.join(tmp, "out"), "safree": True, "svf": True, "lra": True, "task_config": os.path.join(tmp, "task.yaml"),
           "num_inference_steps": args.inference_steps}
    json.dump(cfg, open(os.path.join(tmp, "cfg.json"), "w"))
    jargs = driver.parse_args(["--config", os.path.join(tmp, "cfg.json")])
    verdict = lambda imgs, threshold: (bool(imgs[0].getpixel((0, 0))[0] & 1), 0.5)       # stands where NudeNet sits (out of scope)
    tm = {}
    sync(); t5 = time.perf_counter()
    with open(os.devnull, "w") as null, contextlib.redirect_stdout(null):            # Logger.log prints every line (as the reference does)
        driver.run_job(jargs, pipe, proc, driver.load_task_config(jargs.task_config), eval_func=verdict, prompts_per_batch=P,
                       device=dev, timings=tm)
    sync(); t6 = time.perf_counter()
    nb_ = tm["batches"]
    files = sum(len(fs) for _, _, fs in os.walk(cfg["save_dir"]))
    png_bytes = sum(os.path.getsize(os.path.join(d_, f_)) for d_, _, fs in os.walk(cfg["save_dir"]) for f_ in fs if f_.endswith(".png"))
    steady = (sum(b_["prompts"] for b_ in nb_[1:]) / sum(b_["gpu_s"] for b_ in nb_[1:])) if len(nb_) > 1 else None
    res = {"what": f"driver.run_job over the {total}-row prompt table on 1 GPU, README-default erase_id safree_neg_prompt_rep_threshold_time, "
                   f"{P} prompts per batch, real PNG writes ({files} files, {png_bytes / 1e6:.0f} MB) + stub classifier + logs, host I/O "
                   f"overlapped with the next batch",
           "startup": st, "startup_s": st["startup_s"], "first_batch_s": nb_[0]["gpu_s"], "batches": [b_["prompts"] for b_ in nb_],
           "steady_images_per_sec": steady, "total_s": t6 - t5, "job_images_per_sec": total / (t6 - t5),
           "host_io_busy_s": tm["host_io_s"], "host_io_share_of_job": tm["host_io_s"] / (t6 - t5),
           "whole_process_estimate_s": st["startup_s"] + (t6 - t5), "r5_calibrated_beta_threshold": r5}
    shutil.rmtree(tmp, ignore_errors=True)
    del u, enc, vae, pipe, proc
    torch.cuda.empty_cache()
    return res


def measure_e2e_scheduled(args, dev, proc, P, mine, schedule, lo="f16", timed_batches=2, elide=False):
    """The HEADLINE call with a per-step PRECISION SCHEDULE (round 5): two launch plans over the same weights -- the 16-bit plan
    (`lo`) and the bf16x3 plan -- and `schedule` (SafeDenoiserPipeline.hi_steps forms) says which of the 50 steps run on the
    precise one.  Text encoder bf16x3 throughout (its output feeds the categorical SAFREE decisions), VAE decoder 16-bit (after
    the parity tap).  Which schedules meet the north star's tolerance from token ids is measured by tools/precision_schedule.py
    (profiles/round5_precision_schedule.md) and asserted in tests/test_gpu_e2e_ids.py."""
    from safe_denoiser_amd.clip import CLIPTextModel
    from safe_denoiser_amd.pipeline import SafeDenoiserPipeline, make_scheduler
    from safe_denoiser_amd.unet import UNet2DConditionModel
    from safe_denoiser_amd.vae import AutoencoderKL
    from tests_support.fake_tokenizer import FakeCLIPTokenizer
    dt_lo = torch.float16 if lo == "f16" else torch.bfloat16
    rep = 2 if elide else 3
    u_lo = UNet2DConditionModel(latent_repeat=rep, dtype=dt_lo); u_lo.load_synthetic_on_device(1234, device=dev)
    u_hi = UNet2DConditionModel(latent_repeat=rep, precision="bf16x3"); u_hi.load_synthetic_on_device(1234, device=dev)
    enc = CLIPTextModel(precision="bf16x3"); enc.load_synthetic_on_device(4242, device=dev)
    vae = AutoencoderKL(dtype=_dtype(args)); vae.load_synthetic_on_device(4321, device=dev)
    pipe = SafeDenoiserPipeline(u_lo, make_scheduler(args.scheduler), variant="threshold_time", vae=vae, text_encoder=enc,
                                tokenizer=FakeCLIPTokenizer(), unet_hi=u_hi, precision_schedule=schedule, elide_dead_branch=elide)

    def call(k, n):
        idx = [mine[(k * P + j) % len(mine)] for j in range(P)]
        return pipe([synthetic_prompt(i) for i in idx], num_images_per_prompt=1, guidance_scale=7.5, num_inference_steps=n,
                    negative_prompt=", ".join(NEG_SPACE), negative_prompt_space=NEG_SPACE, height=512, width=512,
                    generator=[torch.Generator(device=dev).manual_seed(1000 + i) for i in idx], repellency_processor=proc,
                    safree_dict=dict(SAFREE), output_type="uint8")
    # warm-up: 12 steps put steps on BOTH plans for every schedule bench.py times (window: t = 914, 831 of 12; first 9: steps 9 .. 11 on
    # the 16-bit plan), so plans, arenas and code objects exist before the clock starts -- a quarter of a full batch's time
    call(0, min(12, args.inference_steps))
    assert 0 < pipe.last_stats["hi_steps"] < min(12, args.inference_steps) or args.inference_steps < 12, pipe.last_stats
    torch.cuda.synchronize(); t0 = time.perf_counter()
    renoise = 0
    for k in range(timed_batches):
        out = call(1 + k, args.inference_steps)
        renoise += pipe.last_stats["renoise_draws"]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / timed_batches
    assert out.dtype == torch.uint8 and tuple(out.shape) == (P, 512, 512, 3)
    res = {"value": P / dt, "unit": "images/sec", "dtype": f"{lo} + bf16x3 on {pipe.last_stats['hi_steps']} of {args.inference_steps} steps",
           "schedule": schedule if not callable(schedule) else "callable", "precise_steps": pipe.last_stats["hi_steps"],
           "workload": "the headline call (README-default end to end, 3 guidance branches)", "prompts_per_batch": P,
           "timed_batches": timed_batches, "ms_per_batch": dt * 1e3, "renoise_draws": renoise,
           "guidance_branches_computed": pipe.last_stats["branches"],
           "window_readbacks_per_batch": pipe.last_stats["window_readbacks"],
           # precise steps at which lra's DISCARDED third branch ran on the 16-bit plan (the two live branches on the precise one)
           "precise_steps_with_discarded_branch_16bit": pipe.last_stats.get("dead_branch_lo_steps", 0)}
    del u_lo, u_hi, enc, vae, pipe
    torch.cuda.empty_cache()
    return res


def measure_parity(args, dev, steps=10):
    """Distance of each engine mode from the engine's own fp32 plan, measured IN THIS RUN: full SD-v1.4 size, 1 prompt,
    CFG 7.5, DDPM, `steps` iterations from identical noise (a tape), every repellency gate firing.  The fp32 plan is the
    stand-in for the reference's fp32 arithmetic (run_nudity.py:277): tests/test_gpu_f32.py pins it to the CPU oracle at
    1.25e-5 on this very loop (profiles/round3_parity.json).  The oracle itself is not touched here."""
    from safe_denoiser_amd.pipeline import SafeDenoiserPipeline, make_scheduler
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
    from safe_denoiser_amd.unet import UNet2DConditionModel
    g = torch.Generator().manual_seed(5)
    E = torch.randn(2, 77, 768, generator=g).to(dev)
    refs = torch.randn(64, 4, 64, 64, generator=g)
    refs = refs / refs.norm(dim=1, keepdim=True)
    tape = torch.randn(4 * steps, 1, 4, 64, 64, generator=g).to(dev)
    path = os.path.join(tempfile.mkdtemp(prefix="sdn_parity_"), "pr.pt")
    torch.save(refs, path)
    proc = thr.get_repellency_method("kernel_fast", torch.zeros(1, device=dev), None, None, 50, 1000, 0.00085, 0.012, n_embed=4,
                                     proj_ref_path=path, cache_proj_ref=True, sigma=3.15, scale=0.33, beta_threshold=1e-6,
                                     beta_threshold_margin=1e9)
    lat, draws = {}, {}
    for name, kw in (("fp32", dict(precision="fp32")), ("bf16x3", dict(precision="bf16x3")), ("f16", dict(dtype=torch.float16)),
                     ("bf16", dict(dtype=torch.bfloat16))):
        u = UNet2DConditionModel(latent_repeat=2, **kw)
        u.load_synthetic_on_device(1234, device=dev)                   # same generator stream: the 16-bit weights are the f32 ones rounded
        cur = [0]

        def noise(p, shape):
            z = tape[cur[0]].reshape(shape)
            cur[0] += 1
            return z
        pipe = SafeDenoiserPipeline(u, make_scheduler("ddpm"), variant="threshold_time")
        lat[name] = pipe(prompt_embeddings=E, num_inference_steps=steps, guidance_scale=7.5, noise_fn=noise,
                         repellency_processor=proc, return_latents=True).double()
        draws[name] = pipe.last_stats["renoise_draws"]
        del u, pipe
        torch.cuda.empty_cache()
    rel = {k: float((v - lat["fp32"]).norm() / lat["fp32"].norm()) for k, v in lat.items() if k != "fp32"}
    return {"what": f"full SD-v1.4, 1 prompt, CFG 7.5, DDPM {steps} steps, tape noise, gate firing: rel L2 of each mode's final "
                    f"latents vs the engine's fp32 plan, measured in this run",
            "truth": "engine fp32 plan (pinned to the CPU fp32 oracle at 1.25e-5 / 1.27e-5 over 10 / 50 steps by tests/test_gpu_f32.py)",
            "loop_rel_l2": rel, "renoise_draws": draws, "north_star_bound": 1e-3,
            "meets_bound": {k: v <= 1e-3 for k, v in rel.items()}}


def measure_sd3(dev, side, P, steps=50, refs_m=515):
    """BASELINE config 4 (SD-v3 medium MMDiT, fp16, repellency_methods_fast_sdv3, M = 515) as a side measurement: `side`
    64 = the reference driver's own 512x512 default (run_nudity_sdv3.py:357-358,500), 128 = the 1024x1024 BASELINE names.
    50 flow-Euler steps, guidance 3.5 (:501-502), synthetic weights / references; its own MFMA roofline from a
    HIP-event-profiled forward."""
    import tempfile as _tf
    from safe_denoiser_amd.mmdit import SD3Transformer2DModel
    from safe_denoiser_amd.pipeline_sd3 import SD3SafeDenoiserPipeline
    from safe_denoiser_amd.repellency import repellency_methods_fast_sdv3 as sd3rep
    from safe_denoiser_amd.schedulers import FlowMatchEulerDiscreteScheduler
    m = SD3Transformer2DModel(sample_size=side)
    m.load_synthetic_on_device(3, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    refs = torch.randn(refs_m, 16, side, side, generator=g, device=dev)
    refs = (refs / refs.norm(dim=1, keepdim=True)).cpu()
    path = os.path.join(_tf.mkdtemp(prefix="sdn_sd3_"), "proj_ref.pt")
    torch.save(refs, path)
    del refs
    proc = sd3rep.get_repellency_method("kernel_fast", torch.zeros(1, device=dev), None, None, 50, 1000, 0.00085, 0.012,
                                        n_embed=4, proj_ref_path=path, cache_proj_ref=True, scale=0.03)
    emb = torch.randn(2 * P, 333, 4096, device=dev)
    pooled = torch.randn(2 * P, 2048, device=dev)
    pipe = SD3SafeDenoiserPipeline(m, FlowMatchEulerDiscreteScheduler())
    dt = 0.0
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pipe(prompt_embeds=emb, pooled_prompt_embeds=pooled, num_inference_steps=steps, guidance_scale=3.5,
                   repellency_processor=proc, generator=[torch.Generator(device=dev).manual_seed(10 + i) for i in range(P)])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert torch.isfinite(out.float()).all()
    # forward profile (HIP events on the launch stream, per kernel label)
    x = torch.randn(2 * P, 16, side, side, device=dev)
    text = m.prepare_text(emb)
    pl = pooled.to(m.dtype).contiguous()
    y = torch.empty_like(x)
    m.forward_into(x, 500.0, text, pl, y)
    m.profile_next()
    m.forward_into(x, 500.0, text, pl, y)
    rows = m.profile_read()
    fwd_ms = sum(r["ms"] for r in rows)
    dom = max(rows, key=lambda r: r["ms"])
    attn = [r for r in rows if r["kernel"].startswith("k_attn")]
    fl, _ = m.flops(2 * P)
    res = {"value": P / dt, "unit": "images/sec", "image": f"{side * 8}x{side * 8}", "dtype": "f16", "prompts_per_batch": P,
           "steps": steps, "window_steps": pipe.last_stats["window_steps"], "ms_per_step": dt / steps * 1e3,
           "mmdit": {"ms_per_forward": fwd_ms, "batch": 2 * P, "tflops": fl / (fwd_ms * 1e-3) / 1e12},
           "roofline": {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["flops"] / (dom["ms"] * 1e-3) / 1e12,
                        "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": dom["flops"] / (dom["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                        "avg_launch_us": dom["ms"] / dom["launches"] * 1e3, "launches_per_forward": dom["launches"]},
           "attention_roofline": {"achieved": sum(r["flops"] for r in attn) / (sum(r["ms"] for r in attn) * 1e-3) / 1e12,
                                  "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s"}}
    res["attention_roofline"]["frac"] = res["attention_roofline"]["achieved"] / PEAK_BF16_TFLOPS
    del m, pipe, proc, emb, pooled, x, y
    torch.cuda.empty_cache()
    return res


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: this (parent) process never touches the GPU; it starts N fresh
    child processes with the torchrun environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), relays rank 0's
    stdout (the JSON line) and the children's stderr, and fails if any rank fails."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SDN_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    bad = []
    while True:                                                     # a rank that dies must not leave the others waiting in
        states = [p.poll() for p in procs]                          # a collective: stop exactly the processes started here
        bad = [(r, rc) for r, rc in enumerate(states) if rc not in (None, 0)]
        if bad or all(rc is not None for rc in states):
            break
        time.sleep(0.2)
    if bad:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        print(f"[bench] ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    reader.join()
    for line in b"".join(chunks).decode().splitlines():            # stdout carries the JSON line only: library chatter on
        dst = sys.stdout if line.startswith("{") else sys.stderr    # rank 0's stdout (gloo's "[Gloo] Rank 0 is connected
        print(line, file=dst)                                       # ..." in the rehearsal mode) goes to stderr
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--launch-check", action="store_true",
                    help="(tests) every rank prints its RANK/WORLD_SIZE as JSON and exits without touching the GPU")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--prompts-per-batch", type=int, default=64)
    ap.add_argument("--inference-steps", type=int, default=50)
    ap.add_argument("--scheduler", default="ddpm", choices=["ddpm", "ddim"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"], help="16-bit storage type of the UNet")
    ap.add_argument("--text-precision", default="bf16x3", choices=["bf16x3", "fp32", "same"],
                    help="precision mode of the CLIP text encoder (same = the UNet's 16-bit storage type)")
    ap.add_argument("--refs", type=int, default=515)
    ap.add_argument("--total-prompts", type=int, default=515)
    ap.add_argument("--fire-fraction", type=float, default=0.5,
                    help="fraction of (prompt, window step) pairs whose repellency gate fires (0 = keep the R5 threshold)")
    ap.add_argument("--workload", default="e2e", choices=["e2e", "latent"],
                    help="e2e = the README-default call end to end (CLIP + SAFREE + 3-branch loop + VAE decode); latent = the round-1/2 "
                         "headline (prompt embeddings in, latents out, 2 CFG branches)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary legs (latent-level b = 2 figure, bf16x3 precision mode, live parity, SD-v3 config 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latent-repeat", action="store_true", help="feed cat([latents] * 2) to a plain UNet plan")
    ap.add_argument("--no-tail-split", action="store_true",
                    help="a batch a few prompts over a multiple of 64 samples (e.g. --prompts-per-batch 65) as ONE forward per step (A/B of pipeline.tail_split)")
    args = ap.parse_args()

    # N > 1 without a launcher: become the launcher BEFORE anything touches the GPU (no re-exec of a GPU process)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if not args.launch_check and os.environ.get("SDN_SHARE_GPU") != "1" and torch.cuda.device_count() < args.gpus:
            sys.exit(f"[bench] --gpus {args.gpus} but this node shows {torch.cuda.device_count()} GPU(s): one process per GPU")
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: start one process per GPU "
                 f"(torchrun --nproc-per-node {args.gpus}) or run `python bench.py --gpus {args.gpus}` without a launcher")

    from safe_denoiser_amd import dist as sdist
    if args.launch_check:                                              # CPU-only rehearsal of the N-rank start-up
        rank, world, local = sdist.init_from_env(backend="gloo")
        total = sdist.sum_over_ranks(float(rank), torch.device("cpu"))
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": total, "local_rank": local}))
        sdist.barrier()
        return
    rank, world, local = sdist.init_from_env()
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    unet, pipe, proc, beta, bcast_ms, comm_ms = build_engine(args, rank, world, dev)
    pipe.tail_split = not args.no_tail_split
    P = args.prompts_per_batch
    e2e = args.workload == "e2e"
    nb = 3 if e2e else 2
    mine = sdist.shard_indices(args.total_prompts, rank, world)        # this rank's prompts of the 515-prompt job
    if e2e:
        neg_prompt = ", ".join(NEG_SPACE)                               # run_nudity.py:345-371 for a safree_neg_prompt erase_id
    else:
        g = torch.Generator().manual_seed(7)
        text_all = torch.randn(args.total_prompts, 77, 768, generator=g)
        uncond = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(8))

    def run(k):
        idx = [mine[(k * P + j) % len(mine)] for j in range(P)]
        gens = [torch.Generator(device=dev).manual_seed(1000 + i) for i in idx]
        if e2e:                                                         # the reference's call (run_nudity.py:439-460), batched
            return pipe([synthetic_prompt(i) for i in idx], num_images_per_prompt=1, guidance_scale=7.5,
                        num_inference_steps=args.inference_steps, negative_prompt=neg_prompt, negative_prompt_space=NEG_SPACE,
                        height=512, width=512, generator=gens, repellency_processor=proc, safree_dict=dict(SAFREE),
                        output_type="uint8")
        E = torch.cat([uncond.expand(P, -1, -1), text_all[idx]]).to(dev)
        return pipe(prompt_embeddings=E, num_inference_steps=args.inference_steps, guidance_scale=7.5, generator=gens,
                    repellency_processor=proc, return_latents=True)

    # ---- gate placement (untimed probe).  The R5-calibrated threshold (beta, computed from the references alone) is kept
    # and reported, but on SYNTHETIC weights the x0 probes sit far from every reference, so that gate never fires and
    # the re-noise draws + sdn_renoise_select would never be timed.  The probe batch runs with the gate shut and records
    # the denominators of every window step; the gate is then put at their median, so that about half of the
    # (prompt, window step) pairs fire in the timed region (the measured fraction is reported).
    r5_beta = beta
    placement = "R5 threshold - margin"
    fire_target = args.fire_fraction
    if fire_target > 0:
        proc.beta_threshold = float("inf")
        pipe.record_den = True
        run(0)
        den_list = pipe.last_stats["denominators"]
        pipe.record_den = False
        dens = torch.cat(den_list).float() if den_list else None
        lo_d, hi_d = (float(dens.min()), float(dens.max())) if dens is not None else (0.0, 0.0)
        if dens is None:                                              # no step inside the window (short --inference-steps runs)
            gate, placement = r5_beta - float(proc.beta_threshold_margin), "R5 threshold - margin (no window step in this run)"
        elif hi_d > lo_d * (1.0 + 1e-6):
            gate, placement = float(torch.quantile(dens, 1.0 - fire_target)), f"den quantile {1.0 - fire_target:.2f} of an untimed probe batch"
        else:
            # degenerate: with RANDOM UNet weights every x0 probe is ~1e3 away from every reference, all weights underflow and
            # every denominator equals epsilon -- no threshold separates the prompts.  Take the worst case instead: the
            # gate below all of them, so EVERY (prompt, window step) pair draws a re-noise tensor and runs the select.
            gate, placement = 0.5 * lo_d, "below every denominator of an untimed probe batch (they all equal epsilon on synthetic weights): every pair fires"
        gate = sdist.broadcast_scalar(gate if rank == 0 else 0.0, dev)
        proc.beta_threshold = gate + float(proc.beta_threshold_margin)          # is_negation = den > beta - margin = gate
        beta = proc.beta_threshold
        sdist.heartbeat("gate placed")
    for k in range(args.warmup):
        run(k)
        sdist.heartbeat(f"warm-up batch {k + 1}/{args.warmup} done")
    sdist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    renoise = 0
    window_pairs = 0
    safree_removed, safree_steps = 0, 0
    for k in range(args.steps):
        out = run(args.warmup + k)
        renoise += pipe.last_stats["renoise_draws"]
        window_pairs += pipe.last_stats["window_steps"] * P
        if e2e and pipe.last_safree is not None:
            safree_removed += sum(pipe.last_safree["n_removed"])
            safree_steps += sum(b_ or 0 for b_ in (pipe.last_safree["beta_adjusted"] or []))
    torch.cuda.synchronize()
    dt_mine = time.perf_counter() - t0                                  # this rank's own clock, before the closing barrier
    tail_split_used = pipe.last_stats.get("tail_split")
    sdist.heartbeat(f"timed region done: {P * args.steps} images in {dt_mine:.1f} s")
    sdist.barrier()
    agg = sdist.throughput_over_ranks(P * args.steps, dt_mine, time.perf_counter() - t0, dev)
    dt, per_rank = agg["window_s"], agg["per_rank"]
    if e2e:
        assert out.dtype == torch.uint8 and tuple(out.shape) == (P, 512, 512, 3)
    else:
        assert torch.isfinite(out).all()

    # ---- live kernel timing (HIP events on the launch stream) of one UNet forward at the benchmark shape ----
    x = torch.randn(nb * P // unet.latent_repeat, 4, 64, 64, device=dev)
    tb = unet.prepare_text(torch.randn(nb * P, 77, 768, device=dev))
    y = torch.empty((nb * P, 4, 64, 64), device=dev)
    unet.forward_into(x, 981.0, tb, y)
    rows_acc = {}
    for _ in range(3):
        unet.profile_next()
        unet.forward_into(x, 981.0, tb, y)
        for r in unet.profile_read():
            a = rows_acc.setdefault(r["kernel"], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            for f in ("launches", "ms", "flops", "bytes"):
                a[f] += r[f]
    # the dominant kernel is picked by SYMBOL: plan labels that launch the same symbol are summed first (the
    # label split k_gemm<10> | k_gemm<10>/rp had let a smaller symbol win), the per-label split is kept underneath
    tname = "F16" if args.dtype == "f16" else "BF16"
    by_sym = {}
    for lab, v in rows_acc.items():
        name, pat = label_symbol(lab, tname)
        e = by_sym.setdefault(name, dict(pattern=pat, labels={}, launches=0, ms=0.0, flops=0.0, bytes=0.0))
        e["labels"][lab] = v
        for f in ("launches", "ms", "flops", "bytes"):
            e[f] += v[f]
    dom = max(by_sym, key=lambda k_: by_sym[k_]["ms"])
    d = by_sym[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    attn = {k_: v for k_, v in rows_acc.items() if k_.startswith("k_attn")}
    attn_tf = sum(v["flops"] for v in attn.values()) / (sum(v["ms"] for v in attn.values()) * 1e-3) / 1e12
    unet_ms = sum(v["ms"] for v in rows_acc.values()) / 3
    total_f, attn_f = unet.flops(nb * P)

    # repellency projection (HBM-bound): algorithmic bytes = one read of proj_ref + x in/out
    xq = torch.randn(P, 4, 64, 64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    proc.conditioning_device(xq)
    e0.record()
    for _ in range(20):
        proc.conditioning_device(xq)
    e1.record(); torch.cuda.synchronize()
    rep_ms = e0.elapsed_time(e1) / 20
    rep_bytes = args.refs * 16384 * 4 + 2 * P * 16384 * 4
    x1q = torch.randn(1, 4, 64, 64, device=dev)                      # the reference's own shape: one query per call
    proc.conditioning_device(x1q)
    e0.record()
    for _ in range(50):
        proc.conditioning_device(x1q)
    e1.record(); torch.cuda.synchronize()
    rep1_ms = e0.elapsed_time(e1) / 50
    rep1_bytes = args.refs * 16384 * 4 + 2 * 16384 * 4

    # where an end-to-end batch spends its time outside the UNet loop: the text front end (tokenise + CLIP encodes + SAFREE
    # projection) and the VAE decode + uint8 tail, each timed on its own (untimed with respect to `value`)
    stages = None
    vae_block = None
    if e2e:
        idx = [mine[j % len(mine)] for j in range(P)]
        prompts = [synthetic_prompt(i) for i in idx]
        torch.cuda.synchronize(); t1 = time.perf_counter()
        E2, _ids, am = pipe._new_encode_prompt(prompts, neg_prompt)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        pipe._safree_prepare(prompts, E2, am, NEG_SPACE, dict(SAFREE))
        torch.cuda.synchronize(); t3 = time.perf_counter()
        zl = torch.randn(P, 4, 64, 64, device=dev) * 0.18215
        pipe.vae.decode_latents_uint8(zl)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        pipe.vae.decode_latents_uint8(zl)
        torch.cuda.synchronize(); t5 = time.perf_counter()
        vae_fl, _ = pipe.vae.flops(1)
        vae_ms = (t5 - t4) * 1e3 / P
        batch_ms = dt / args.steps * 1e3
        stages = {"batch_ms": batch_ms, "encode_prompts_ms": (t2 - t1) * 1e3, "safree_masked_encodes_and_projection_ms": (t3 - t2) * 1e3,
                  "vae_decode_uint8_ms": (t5 - t4) * 1e3,
                  "unet_loop_and_rest_ms": batch_ms - (t2 - t1) * 1e3 - (t3 - t2) * 1e3 - (t5 - t4) * 1e3,
                  "note": "front end and decode re-timed after the timed region on one batch; the remainder is the 50-iteration loop"}
        vae_block = {"ms_per_image": vae_ms, "tflop_per_image": vae_fl / 1e12, "tflops": vae_fl / (vae_ms * 1e-3) / 1e12,
                     "note": f"decode_latents + uint8 conversion of {P} images (inside the timed region of `value`)"}
        del zl

    # HBM bytes per launch and matrix-pipe utilisation per symbol from the PMC passes (tools/pmc_traffic.py, tools/pmc_mfma.py;
    # collected in separate rocprofv3 --pmc runs, which cannot be combined with timing).  A record carries the sha256 of the
    # libsdn.so it was collected on: a record made on another build is NOT reported (null with the reason).
    import hashlib
    import re
    import safe_denoiser_amd as sda
    lib_sha = hashlib.sha256(open(sda.lib_path(), "rb").read()).hexdigest()
    rec_t, traffic_src = _pmc_records("traffic", lib_sha)
    rec_m, mfma_src = _pmc_records("mfma_util", lib_sha)

    def pmc_for(pat):
        out = dict(traffic=None, mfma_busy=None, detail=None)
        if pat is None:
            return out
        rx = re.compile(pat)
        if rec_t is not None:
            hit = [r_ for k_, r_ in rec_t.items() if k_ != "__meta__" and rx.search(k_)]
            if hit:
                out["traffic"] = sum(r_["hbm_bytes_per_launch"] * r_["launches"] for r_ in hit) / sum(r_["launches"] for r_ in hit)
        if rec_m is not None:
            hit = [r_ for k_, r_ in rec_m.items() if k_ != "__meta__" and rx.search(k_)]
            wsum = sum(r_["time_share_ms"] for r_ in hit)
            if hit and wsum > 0:
                out["mfma_busy"] = sum(r_["mfma_busy"] * r_["time_share_ms"] for r_ in hit) / wsum
                out["detail"] = {"clock_ghz": sum((r_["clock_ghz_sq"] or 0.0) * r_["time_share_ms"] for r_ in hit) / wsum,
                                 "lds_issue_stall_share_of_wave_cycles": sum((r_["lds_issue_stall_share_of_wave_cycles"] or 0.0) * r_["time_share_ms"] for r_ in hit) / wsum,
                                 "issue_stall_share_of_wave_cycles": sum((r_.get("issue_stall_share_of_wave_cycles") or 0.0) * r_["time_share_ms"] for r_ in hit) / wsum}
        return out

    def sym_row(name):
        e = by_sym[name]
        pm = pmc_for(e["pattern"])
        tf = e["flops"] / (e["ms"] * 1e-3) / 1e12 if e["flops"] else None
        return {"share_of_unet_time": e["ms"] / 3 / unet_ms, "launches_per_forward": e["launches"] // 3,
                "avg_launch_us": e["ms"] / e["launches"] * 1e3, "tflops": tf, "frac": None if tf is None else tf / PEAK_BF16_TFLOPS,
                "algorithmic_bytes_per_launch": e["bytes"] / e["launches"], "traffic": pm["traffic"], "mfma_busy": pm["mfma_busy"],
                **({"mfma_busy_detail": pm["detail"]} if pm["detail"] else {}),
                "labels": {lab: {"launches_per_forward": v["launches"] // 3, "ms_per_forward": v["ms"] / 3,
                                 "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else None}
                           for lab, v in sorted(e["labels"].items(), key=lambda kv: -kv[1]["ms"])}}
    top5 = sorted(by_sym, key=lambda k_: -by_sym[k_]["ms"])[:5]
    by_symbol = {name: sym_row(name) for name in top5}
    dom_row = by_symbol[dom]
    traffic, mfma_busy, mfma_extra = dom_row["traffic"], dom_row["mfma_busy"], dom_row.get("mfma_busy_detail")
    n_img = world * P * args.steps
    value = n_img / dt
    by_tf = {k_: v["flops"] / (v["ms"] * 1e-3) / 1e12 for k_, v in sorted(rows_acc.items(), key=lambda kv: -kv[1]["ms"])[:5]}
    if e2e:
        workload = (f"README-default end to end (safree = svf = lra = true, safree_neg_prompt_config.json:26-28): token ids -> CLIP -> "
                    f"SAFREE projection -> {args.scheduler.upper()} {args.inference_steps} steps x UNet on 3 guidance branches with "
                    f"per-prompt text switching + safe_denoiser.yaml repellency (kernel_fast, M={args.refs}, sigma 3.15, scale .33, "
                    f"margin 1.6, window 780<=t<=1000) -> VAE decode -> uint8 512x512 images; SD-v1.4 UNet / CLIP / VAE with random "
                    f"weights, {args.total_prompts}-prompt job sharded r::{world}, CFG 7.5, 120.5 TFLOP of UNet work per image")
    else:
        workload = (f"SD-v1.4 UNet (random weights) + safe_denoiser.yaml repellency (kernel_fast, M={args.refs}, sigma 3.15, scale "
                    f".33, margin 1.6, window 780<=t<=1000), {args.total_prompts}-prompt job sharded r::{world}, CFG 7.5 (2 "
                    f"branches), {args.scheduler.upper()} {args.inference_steps} steps, prompt embeddings in, 64x64x4 latents out")
    line = {
        "metric": "images/sec (512x512, 50 steps, SD-v1.4 + repellency)", "value": value, "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload, "guidance_branches": nb, "text_encoder_precision": args.text_precision,
                   "prompts_per_batch": P, "latent_repeat": unet.latent_repeat, "tail_split": tail_split_used, "images_timed": n_img, "beta_threshold": beta,
                   "renoise_draws_rank0": renoise,
                   "gate": {"r5_calibrated_beta_threshold": r5_beta, "margin": float(proc.beta_threshold_margin),
                            "placement": placement,
                            "fired_fraction_rank0": renoise / max(window_pairs, 1)},
                   **({"safree": {"trigger_tokens_removed_rank0": safree_removed,
                                  "mean_projected_text_steps_per_prompt": safree_steps / max(P * args.steps, 1)}} if e2e else {}),
                   "parallelism": f"prompt-shard x{world}",
                   "per_rank_images_per_sec": per_rank, "proj_ref_broadcast_ms": bcast_ms, "communicator_setup_ms": comm_ms,
                   **({"rehearsal": "SDN_SHARE_GPU=1: ranks share the visible GPU(s) and talk over gloo -- a functional check of "
                                    "the N > 1 path, NOT a scaling measurement"} if os.environ.get("SDN_SHARE_GPU") == "1" else {})},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "mfma_busy": mfma_busy, "mfma_busy_source": mfma_src, **({"mfma_busy_detail": mfma_extra} if mfma_extra else {}),
                     "algorithmic_bytes_per_launch": d["bytes"] / d["launches"], "kernel": dom,
                     "kernel_selection": "plan labels summed by the kernel symbol they launch; the symbol with the largest share of the "
                                         "HIP-event-timed forward (3 profiled forwards at the benchmark batch)",
                     "labels": dom_row["labels"],
                     "launches_per_forward": d["launches"] // 3, "avg_launch_us": d["ms"] / d["launches"] * 1e3,
                     "share_of_unet_time": d["ms"] / 3 / unet_ms,
                     "by_symbol": by_symbol},
        "attention_roofline": {"achieved": attn_tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": attn_tf / PEAK_BF16_TFLOPS,
                               # diagnostic only: MFMA work actually ISSUED by the flash kernel, which pads the head dim to
                               # the 32x32x16 tile (QK^T: d -> 16-multiple, PV: d + ones column -> 32-multiple); `achieved`
                               # above counts the algorithmic 4*B*H*Nq*Nk*d only
                               "mfma_issued_tflops": {k_: (v["flops"] / (v["ms"] * 1e-3) / 1e12) * _attn_pad(k_)
                                                      for k_, v in attn.items()},
                               "kernels": {k_: {"avg_launch_us": v["ms"] / v["launches"] * 1e3,
                                                "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12} for k_, v in attn.items()}},
        "unet": {"ms_per_forward": unet_ms, "batch": nb * P, "tflops": total_f / (unet_ms * 1e-3) / 1e12,
                 "attention_core_share_of_flops": attn_f / total_f,
                 "by_kernel_ms": {k_: v["ms"] / 3 for k_, v in sorted(rows_acc.items(), key=lambda kv: -kv[1]["ms"])},
                 "by_kernel_tflops": by_tf},
        "repellency_roofline": {"bound": "hbm", "achieved": rep_bytes / (rep_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                                "unit": "GB/s", "frac": rep_bytes / (rep_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                "us_per_call": rep_ms * 1e3, "queries": P,
                                "single_query": {"us_per_call": rep1_ms * 1e3,
                                                 "achieved": rep1_bytes / (rep1_ms * 1e-3) / 1e9,
                                                 "frac": rep1_bytes / (rep1_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}},
    }
    if stages is not None:
        line["stages"] = stages
    if vae_block is not None:
        line["vae_decode"] = vae_block
    if rank == 0 and world == 1 and not args.no_extras:
        del unet, pipe
        torch.cuda.empty_cache()
        sdist.heartbeat("secondary legs")
        if e2e:          # the round-1/2 headline, same gate: what the 16-bit engine does on the latent-level workload ...
            line["latent_b2"] = measure_latent(args, dev, proc, P, steps_timed=1)
            line["e2e_dead_branch_elided"] = measure_e2e_elided(args, dev, proc, P, mine)
        # ... and the mode that meets the north star's 1e-3 on that same workload (fp32 storage, split-operand contractions)
        line["precision_mode_bf16x3"] = measure_latent(args, dev, proc, min(P, 32), precision="bf16x3", steps_timed=1)
        if "latent_b2" in line:
            line["precision_mode_bf16x3"]["relative_to_16bit_engine_same_workload"] = \
                line["precision_mode_bf16x3"]["value"] / line["latent_b2"]["value"]
        # the headline call itself in the mode that meets the tolerance from token ids (VERDICT r3 next #1c)
        line["e2e_bf16x3"] = measure_e2e_precision(args, dev, proc, min(P, 64), mine, timed_batches=1)      # (B = 192 samples per forward: 3.3 % per image over 96)
        line["e2e_bf16x3"]["relative_to_headline"] = line["e2e_bf16x3"]["value"] / value
        # ... and the mode that meets it at the lowest cost (round 5): the fp16 plan with the bf16x3 plan on the 11 steps of the
        # repellency window only.  tools/precision_schedule.py: a 16-bit step inside the window moves the final latents by 2.6e-4 ...
        # 3.0e-3, one outside it by 2e-6 ... 3e-5; tests/test_gpu_e2e_ids.py asserts ids -> latents <= 5e-4 with every SAFREE / gate
        # decision and draw count equal to the pure-fp32 chain's.
        line["e2e_scheduled"] = measure_e2e_scheduled(args, dev, proc, min(P, 64), mine, {"window": True})
        line["e2e_scheduled"]["relative_to_headline"] = line["e2e_scheduled"]["value"] / value
        line["e2e_scheduled"]["ids_to_latents_rel_l2_vs_fp32_chain"] = ("1.0e-4 max over 8 prompts, all decisions / draw counts equal "
                                                                         "(profiles/round5_precision_schedule.md, round5_e2e_ids.json)")
        line["value_at_north_star_tolerance"] = line["e2e_scheduled"]["value"]
        line["e2e_scheduled_dead_branch_elided"] = dict(
            measure_e2e_scheduled(args, dev, proc, min(P, 64), mine, {"window": True}, timed_batches=1, elide=True),
            note="the scheduled mode with lra's discarded third branch not computed (bit-identical latents, 2/3 of the UNet work): a labelled "
                 "secondary, NOT value_at_north_star_tolerance")
        # the rest of the headroom, as a labelled frontier point (NOT value_at_north_star_tolerance, which keeps the 10 x margin): the
        # precise plan on the first 9 steps only = 4.1-4.3e-4 from the fp32 chain on three weight seeds, every decision / draw count equal
        # (profiles/round5_precision_schedule_first_n_seed*.md; asserted <= 1e-3 in tests/test_gpu_e2e_ids.py)
        line["e2e_scheduled_first_9"] = dict(
            measure_e2e_scheduled(args, dev, proc, min(P, 64), mine, {"first": 9}, timed_batches=1),
            ids_to_latents_rel_l2_vs_fp32_plans="4.1e-4 ... 4.3e-4 max over 8 prompts on 3 weight seeds (north star: 1e-3), all decisions / draw "
                                                "counts equal",
            note="frontier point: 2.3 x inside the bound where e2e_scheduled (the window schedule) is 10 x inside it")
        line["value_at_north_star_tolerance_mode"] = ("e2e_scheduled (fp16 plan + bf16x3 plan inside the repellency window, there on the two live "
                                                      "guidance branches -- lra's discarded third branch stays on the fp16 plan; text encoder bf16x3)")
        line["parity"] = measure_parity(args, dev)
        line["job_515"] = measure_job(args, dev, P, beta)
        line["job_515"]["steady_vs_value"] = (line["job_515"]["steady_images_per_sec"] or 0.0) / value
        # prompts per batch from the sweep in profiles/round3_sd3_batch_sweep.txt (512^2: P = 4 / 8 / 16 / 32 -> 8.6 / 9.8 / 10.6 / 10.8
        # images/sec over 20 steps; 1024^2: P = 2 / 4 / 8 -> 2.19 / 2.44 / 2.51): the knee, not the last per cent
        line["sd3_config4"] = {"512x512": measure_sd3(dev, 64, 16), "1024x1024": measure_sd3(dev, 128, 8)}
    import glob as _glob
    ppaths = sorted(_glob.glob(os.path.join(ROOT, "profiles", "round*_parity.json")),
                    key=lambda f: int("".join(ch for ch in os.path.basename(f).split("_")[0] if ch.isdigit()) or 0))
    if ppaths:                                      # the GPU suite's NEWEST record vs the CPU oracle (tests/test_gpu_f32.py), committed
        line.setdefault("parity", {})["suite_record_vs_cpu_oracle"] = dict(json.load(open(ppaths[-1]))["modes"],
                                                                           source=f"profiles/{os.path.basename(ppaths[-1])}")
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            sdist.heartbeat("cpu baseline")
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line))


if __name__ == "__main__":
    main()
