"""Which steps of the 50-step loop have to run on the precise plan?  (VERDICT r4 next #1.)

The reference runs text encoder, SAFREE and UNet in fp32 (run_nudity.py:277); north star: final latents <= 1e-3 rel.  The engine
meets that with its bf16x3 plan at 1/3 of the 16-bit engine's speed and misses it 4.7x with fp16 everywhere.  This tool measures,
at FULL SD-v1.4 size on the README-default call of tests/test_gpu_e2e_ids.py (8 prompts in one batch, safree + svf + lra, three
guidance branches, 50 DDPM steps, the repellency gate firing at all 11 window steps, the same per-prompt noise tapes), what a
PER-STEP precision schedule buys: `SafeDenoiserPipeline(unet=<fp16 plan>, unet_hi=<bf16x3 plan>, precision_schedule=...)`.

  1. truth          = the engine's fp32 plans (UNet + text encoder; 3e-5 from the pure-fp32 oracle chain, profiles/round4_e2e_ids.json)
  2. sensitivities  = for every step k: bf16x3 everywhere EXCEPT step k (fp16 there) -> distance of the final latents from the
                      all-bf16x3 run = what ONE fp16 step at position k costs; and the converse (fp16 everywhere except k)
  3. schedules      = bf16x3 on the first K / the last K / the window (+ last K) / the K most sensitive steps by (2)

Text encoder: bf16x3 in every arm (its output feeds the SAFREE decisions, which are categorical).  Cost model: a step on the precise
plan costs `--ratio` (default 2.95: 469 / 159 ms per forward at B = 192) times a 16-bit step.

    python tools/precision_schedule.py [--seed 1234] [--quick] [--out gpurun_out/round5_precision_schedule]
Writes <out>.json and <out>.md."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from safe_denoiser_amd.clip import CLIPTextModel  # noqa: E402
from safe_denoiser_amd.pipeline import SafeDenoiserPipeline  # noqa: E402
from safe_denoiser_amd.repellency import repellency_methods_threshold as thr  # noqa: E402
from safe_denoiser_amd.schedulers import DDIMScheduler, DDPMScheduler  # noqa: E402
from safe_denoiser_amd.unet import UNet2DConditionModel  # noqa: E402
from tests_support.fake_tokenizer import FakeCLIPTokenizer  # noqa: E402
from tests_support.e2e_case import NEG_SPACE, PARAMS, PROMPTS, SF, STEPS, Tapes, make_refs  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--ratio", type=float, default=2.95)
    ap.add_argument("--quick", action="store_true", help="every 5th sensitivity arm only")
    ap.add_argument("--verify", action="store_true", help="another weight seed: the two baselines and the candidate schedules only")
    ap.add_argument("--arms", default="", help='with --verify: extra "first N" arms, e.g. --arms 7,8,9')
    ap.add_argument("--lo", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--scheduler", default="ddpm", choices=["ddpm", "ddim"], help="ddpm = the reference's live scheduler; ddim = the one BASELINE names")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "round5_precision_schedule"))
    a = ap.parse_args()
    Sched = DDPMScheduler if a.scheduler == "ddpm" else DDIMScheduler
    t00 = time.time()
    P = len(PROMPTS)
    tok = FakeCLIPTokenizer()
    usd = UNet2DConditionModel(text_len=77).synthetic_state_dict(a.seed)
    csd = CLIPTextModel().synthetic_state_dict(31)
    refs = make_refs()
    path = "/tmp/precision_schedule_refs.pt"
    torch.save(refs, path)
    proc = thr.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085, 0.012, n_embed=4,
                                     proj_ref_path=path, cache_proj_ref=True, **PARAMS)
    shape = (1, 4, 64, 64)

    def unet(**kw):
        u = UNet2DConditionModel(text_len=77, latent_repeat=3, **kw)
        u.load_state_dict(usd)
        return u

    def clip(**kw):
        e = CLIPTextModel(**kw)
        e.load_state_dict(csd)
        return e

    def call(pipe):
        tapes = Tapes(P, shape, 3 * STEPS + 4, seed=77)
        t0 = time.time()
        lat = pipe(PROMPTS, num_inference_steps=STEPS, guidance_scale=7.5, negative_prompt=", ".join(NEG_SPACE),
                   negative_prompt_space=NEG_SPACE, repellency_processor=proc, safree_dict=SF, noise_fn=tapes, return_latents=True)
        torch.cuda.synchronize()
        return lat, dict(draws=pipe.last_stats["renoise_draws"], hi_steps=pipe.last_stats["hi_steps"], cursors=list(tapes.cur),
                         beta_adjusted=list(pipe.last_safree["beta_adjusted"]), n_removed=list(pipe.last_safree["n_removed"]),
                         seconds=round(time.time() - t0, 2))

    # ---- truth: the engine's fp32 plans
    u32, e32 = unet(precision="fp32"), clip(precision="fp32")
    truth, st_truth = call(SafeDenoiserPipeline(u32, Sched(), variant="threshold_time", text_encoder=e32, tokenizer=tok))
    del u32, e32
    torch.cuda.empty_cache()
    print(f"[{time.time() - t00:6.0f} s] truth (fp32 plans): {st_truth}", flush=True)

    u_lo = unet(dtype=torch.float16 if a.lo == "f16" else torch.bfloat16)
    u_hi = unet(precision="bf16x3")
    e_hi = clip(precision="bf16x3")
    sch = Sched()
    sch.set_timesteps(STEPS)
    timesteps = [int(t) for t in sch.timesteps]

    def arm(name, spec):
        pipe = SafeDenoiserPipeline(u_lo, Sched(), variant="threshold_time", text_encoder=e_hi, tokenizer=tok, unet_hi=u_hi,
                                    precision_schedule=spec)
        lat, st = call(pipe)
        per = [rel(lat[p], truth[p]) for p in range(P)]
        n_hi = st["hi_steps"]
        r = dict(name=name, hi_steps=n_hi, cost_vs_16bit=round((n_hi * a.ratio + (STEPS - n_hi)) / STEPS, 3), worst=max(per),
                 per_prompt=per, decisions_equal=(st["beta_adjusted"] == st_truth["beta_adjusted"] and st["n_removed"] == st_truth["n_removed"]),
                 draws_equal=(st["draws"] == st_truth["draws"] and st["cursors"] == st_truth["cursors"]), seconds=st["seconds"])
        print(f"[{time.time() - t00:6.0f} s] {name:34s} hi {n_hi:2d}/50  cost x{r['cost_vs_16bit']:.2f}  worst {r['worst']:.2e}  "
              f"decisions {'=' if r['decisions_equal'] else 'DIFFER'} draws {'=' if r['draws_equal'] else 'DIFFER'}  ({st['seconds']} s)", flush=True)
        return r, lat

    res = dict(what=__doc__.split("\n\n")[0], seed=a.seed, lo=a.lo, scheduler=a.scheduler, ratio=a.ratio, timesteps=timesteps, truth=st_truth, arms=[])
    r_all, lat_all = arm("all bf16x3", "all")
    r_none, lat_none = arm(f"all {a.lo}", "none")
    res["arms"] += [r_all, r_none]

    if a.verify:
        for name, spec in (("window (11)", {"window": True}), ("first 10", {"first": 10}), ("first 15", {"first": 15}),
                           ("window (11) + last 5", {"window": True, "last": 5})) + tuple(
                               (f"first {int(k)}", {"first": int(k)}) for k in a.arms.split(",") if k.strip()):
            res["arms"].append(arm(name, spec)[0])
        res["wall_s"] = round(time.time() - t00, 1)
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(res, open(a.out + ".json", "w"), indent=1)
        with open(a.out + ".md", "w") as f:
            f.write(f"# Precision schedules, verification arms (weight seed {a.seed}; 16-bit plan = {a.lo}; scheduler = {a.scheduler})\n\n| arm | precise steps | cost | worst rel L2 | "
                    f"decisions / draws equal |\n|---|---|---|---|---|\n")
            for r in res["arms"]:
                f.write(f"| {r['name']} | {r['hi_steps']} | x{r['cost_vs_16bit']:.2f} | {r['worst']:.2e} | "
                        f"{'yes' if r['decisions_equal'] and r['draws_equal'] else 'NO'} |\n")
        print("wrote", a.out + ".json/.md", f"in {res['wall_s']} s")
        return

    # ---- sensitivities
    ks = list(range(0, STEPS, 5 if a.quick else 1))
    sens, conv = {}, {}
    for k in ks:
        r, lat = arm(f"bf16x3 except step {k} (t={timesteps[k]})", [i != k for i in range(STEPS)])
        r["from_all_hi"] = max(rel(lat[p], lat_all[p]) for p in range(P))
        sens[k] = r["from_all_hi"]
        res["arms"].append(r)
    for k in (ks if not a.quick else []):
        r, lat = arm(f"{a.lo} except step {k} (t={timesteps[k]})", {"steps": [k]})
        r["gain_vs_all_lo"] = r_none["worst"] - r["worst"]
        conv[k] = r["gain_vs_all_lo"]
        res["arms"].append(r)
    res["sensitivity_one_lo_step"] = sens
    res["gain_one_hi_step"] = conv

    # ---- schedules
    order = sorted(sens, key=lambda k: -sens[k])
    for K in (5, 10, 11, 15, 20, 25, 30, 35, 40, 45):
        res["arms"].append(arm(f"first {K}", {"first": K})[0])
    for K in (5, 10, 15, 20, 25, 30, 40):
        res["arms"].append(arm(f"last {K}", {"last": K})[0])
    for K in (5, 10, 15, 20):
        res["arms"].append(arm(f"window (11) + last {K}", {"window": True, "last": K})[0])
    if not a.quick:
        for K in (5, 10, 15, 17, 20, 25, 30, 35, 40):
            res["arms"].append(arm(f"the {K} most sensitive", {"steps": order[:K]})[0])
    res["most_sensitive_order"] = order
    passing = [r for r in res["arms"] if r["worst"] <= 5e-4 and r["decisions_equal"] and r["draws_equal"]]
    passing1 = [r for r in res["arms"] if r["worst"] <= 1e-3 and r["decisions_equal"] and r["draws_equal"]]
    res["cheapest_within_5e-4"] = min(passing, key=lambda r: r["cost_vs_16bit"]) if passing else None
    res["cheapest_within_1e-3"] = min(passing1, key=lambda r: r["cost_vs_16bit"]) if passing1 else None
    res["wall_s"] = round(time.time() - t00, 1)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out + ".json", "w"), indent=1)
    with open(a.out + ".md", "w") as f:
        f.write(f"# Per-step precision schedule sweep (weights seed {a.seed}; 16-bit plan = {a.lo}; precise plan = bf16x3 at x{a.ratio} per step)\n\n")
        f.write("Source: `tools/precision_schedule.py`; truth = the engine's fp32 plans; worst = max over the 8 prompts of the final-latents rel L2.\n\n")
        f.write("| arm | precise steps | cost vs all-16-bit | worst rel L2 | <= 5e-4 | <= 1e-3 | decisions / draws equal |\n|---|---|---|---|---|---|---|\n")
        for r in res["arms"]:
            if "except step" in r["name"]:
                continue
            f.write(f"| {r['name']} | {r['hi_steps']} | x{r['cost_vs_16bit']:.2f} | {r['worst']:.2e} | {'yes' if r['worst'] <= 5e-4 else 'no'} | "
                    f"{'yes' if r['worst'] <= 1e-3 else 'no'} | {'yes' if r['decisions_equal'] and r['draws_equal'] else 'NO'} |\n")
        f.write("\n## What ONE 16-bit step costs (bf16x3 everywhere else), by position\n\n| step | t | distance from the all-bf16x3 run |\n|---|---|---|\n")
        for k in ks:
            f.write(f"| {k} | {timesteps[k]} | {sens[k]:.2e} |\n")
        f.write(f"\nSum of the single-step distances: {sum(sens.values()):.2e} (all-{a.lo} run: {r_none['worst']:.2e} from the truth).\n")
        f.write(f"\nCheapest arm within 5e-4: {res['cheapest_within_5e-4'] and res['cheapest_within_5e-4']['name']}; within 1e-3: "
                f"{res['cheapest_within_1e-3'] and res['cheapest_within_1e-3']['name']}.\n")
    print("wrote", a.out + ".json/.md", f"in {res['wall_s']} s")


if __name__ == "__main__":
    main()
