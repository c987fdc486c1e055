#!/usr/bin/env python3
"""Capture golden vectors from the reference's own repellency modules.

Runs ONLY in the build container (needs /root/reference on PYTHONPATH):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python /root/repo/tests/golden/make_golden.py

Writes tests/golden/repellency_golden.npz (inputs + expected outputs only; no
reference source travels).  Case list = SURVEY.md section 8c G1..G8.
Each case stores: the inputs (x, refs, params) and what the reference returned.
"""
import importlib
import os
import sys
import tempfile

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "repellency_golden.npz")
MODS = {
    "threshold": "repellency.repellency_methods_threshold",
    "fast": "repellency.repellency_methods_fast",
    "fast_sdv3": "repellency.repellency_methods_fast_sdv3",
}


def chan_norm_refs(m, c, hw, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    r = torch.randn(m, c, hw, hw, generator=g)
    r = r / torch.norm(r, dim=1, keepdim=True)
    return r.to(dtype)


def make_proc(flavour, method, refs, tmpdir, **params):
    mod = importlib.import_module(MODS[flavour])
    path = os.path.join(tmpdir, f"pr_{flavour}_{method}_{abs(hash(str(params))) % 10**8}.pt")
    torch.save(refs, path)
    ref_data = torch.zeros(1, dtype=refs.dtype)           # only .device/.dtype are read once cached
    return mod.get_repellency_method(method, ref_data, None, None, 50, 1000, 0.00085, 0.012,
                                     n_embed=4, proj_ref_path=path, cache_proj_ref=True, **params)


def main_sdv3_full():
    """G8-style FULL-SIZE cases of BASELINE config 4's projection (repellency_methods_fast_sdv3.py:229-271): M = 515 references of
    the SD-v3 latent shape [16, 64, 64] (D = 65 536: the reference driver's own 512 x 512 default, run_nudity_sdv3.py:357-358).
    Stored as head / tail / float64 sum / L2 + the largest change, like G8: written to repellency_golden_sdv3_full.npz (the
    G1-G8 file is left as it is)."""
    torch.set_num_threads(8)
    store, meta = {}, []

    def put(name, **arrs):
        for k, v in arrs.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            store[f"{name}/{k}"] = np.asarray(v)
        meta.append(name)

    with tempfile.TemporaryDirectory() as td:
        m, c, hw = 515, 16, 64
        refs = chan_norm_refs(m, c, hw, 0)
        proc = make_proc("fast_sdv3", "kernel_fast", refs, td, scale=0.03)
        g = torch.Generator().manual_seed(2000)
        noise = torch.randn(1, c, hw, hw, generator=g)
        cases = {
            # an un-normalised query sitting next to reference 7 (after the channel norm its distance is ~1: weights of order 1)
            "near": 3.0 * (refs[7:8].clone() + 0.0005 * noise),
            # between two references
            "between": 2.0 * (0.5 * refs[100:101] + 0.5 * refs[300:301] + 0.0003 * noise),
            # a random query: every distance ~ 90, every weight underflows against epsilon -> the output is the input
            "far": noise.clone(),
            # fp16 input (the SD-v3 pipelines hand over fp16 latents): cast to the references' dtype first
            "near_f16": (3.0 * (refs[7:8].clone() + 0.0005 * noise)).half(),
        }
        for tag, x in cases.items():
            xin = x.clone()
            out = proc.conditioning(xin)
            ox = out["x_0_hat"].reshape(-1)
            put(f"G8_sdv3_full_{tag}", seed_refs=0, seed_noise=2000, m=m, scale=0.03, epsilon=1e-8,
                head=ox[:16].float(), tail=ox[-16:].float(), sum64=float(ox.double().sum()), l2_64=float(ox.double().norm()),
                max_abs_delta=float((out["x_0_hat"].float() - x.float()).abs().max()),
                delta_l2=float((out["x_0_hat"].double() - x.double()).norm()), out_is_f32=int(out["x_0_hat"].dtype == torch.float32))
    store["__cases__"] = np.array(meta)
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "repellency_golden_sdv3_full.npz")
    np.savez_compressed(out_path, **store)
    print(f"wrote {out_path}: {meta}")
    for n in meta:
        print(n, "max_abs_delta", float(store[n + "/max_abs_delta"]), "delta_l2", float(store[n + "/delta_l2"]))


def main():
    torch.set_num_threads(4)
    store = {}
    meta = []

    def put(name, **arrs):
        for k, v in arrs.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().float().numpy() if v.dtype in (torch.float16, torch.bfloat16) else v.detach().cpu().numpy()
            store[f"{name}/{k}"] = np.asarray(v)
        meta.append(name)

    with tempfile.TemporaryDirectory() as td:
        # ---------------- G1/G2: threshold kernel_fast ----------------
        for (m, c, hw) in [(1, 4, 4), (7, 4, 8), (33, 16, 4), (40, 4, 8)]:
            for seed in (0, 1, 2):
                refs = chan_norm_refs(m, c, hw, 100 + seed)
                g = torch.Generator().manual_seed(seed)
                x = torch.randn(1, c, hw, hw, generator=g)
                if seed == 2:                                  # a query sitting close to one ref
                    x = refs[:1].clone() * 1.05 + 0.01 * x
                sigma, scale, margin = 3.15, 0.33, 1.6
                # probe the denominator first, then straddle it
                p0 = make_proc("threshold", "kernel_fast", refs, td, sigma=sigma, scale=scale,
                               beta_threshold=1.0, beta_threshold_margin=0.0)
                den = p0.conditioning(x.clone(), beta_threshold=True)["mean_x_0_hat"]["denominator"]
                for tag, thr in (("lo", den * 0.5 + margin), ("hi", den * 2.0 + margin + 1e-3)):
                    proc = make_proc("threshold", "kernel_fast", refs, td, sigma=sigma, scale=scale,
                                     beta_threshold=float(thr), beta_threshold_margin=margin)
                    xin = x.clone()
                    out = proc.conditioning(xin, beta_threshold=True)
                    assert out["x_0_hat"] is xin
                    put(f"G1_m{m}_c{c}_h{hw}_s{seed}_{tag}", x=x, refs=refs, sigma=sigma, scale=scale,
                        epsilon=1e-8, beta_threshold=float(thr), margin=margin, out_x=out["x_0_hat"],
                        den=out["mean_x_0_hat"]["denominator"], isneg=int(out["is_negation"]),
                        item=out["mean_x_0_hat"]["negative_score_item"])
                proc = make_proc("threshold", "kernel_fast", refs, td, sigma=sigma, scale=scale,
                                 beta_threshold=1.0, beta_threshold_margin=0.0)
                xin = x.clone()
                out = proc.conditioning(xin, beta_threshold=False)
                put(f"G2_m{m}_c{c}_h{hw}_s{seed}", x=x, refs=refs, sigma=sigma, scale=scale, epsilon=1e-8,
                    out_neg=out["x_0_hat"], mutated_x=xin, isneg=int(out["is_negation"]))

        # ---------------- G3: fast ----------------
        for (m, c, hw) in [(7, 4, 8), (33, 4, 4)]:
            for seed in (0, 1):
                refs = chan_norm_refs(m, c, hw, 200 + seed)
                g = torch.Generator().manual_seed(10 + seed)
                # keep distances small so that sigma=1 weights do not underflow everywhere
                x = refs[seed:seed + 1].clone() + 0.05 * torch.randn(1, c, hw, hw, generator=g)
                proc = make_proc("fast", "kernel_fast", refs, td, sigma=3.15, scale=0.33)
                assert not isinstance(getattr(proc, "sigma", None), (int, float))   # YAML sigma ignored
                xin = x.clone()
                out = proc.conditioning(xin)
                put(f"G3a_m{m}_h{hw}_s{seed}", x=x, refs=refs, scale=0.33, epsilon=1e-8,
                    out_x=out["x_0_hat"], alias=int(out["x_0_hat"] is xin), item=out["mean_x_0_hat"])
                xin = x.clone()
                out = proc.conditioning(xin, guidance_scale=1.0)
                put(f"G3b_m{m}_h{hw}_s{seed}", x=x, refs=refs, scale=0.33, epsilon=1e-8,
                    out_neg=out["x_0_hat"], mutated_x=xin)
                xh = x.clone().half()
                out = proc.conditioning(xh)
                put(f"G3c_m{m}_h{hw}_s{seed}", x=xh.float(), refs=refs, scale=0.33, epsilon=1e-8,
                    out_x=out["x_0_hat"], out_is_f32=int(out["x_0_hat"].dtype == torch.float32),
                    input_unchanged=int(torch.equal(xh, x.clone().half())))

        # ---------------- G4: fast_sdv3 (channel-normalised query) ----------------
        for seed in (0, 1):
            m, c, hw = 9, 16, 4
            refs = chan_norm_refs(m, c, hw, 300 + seed)
            g = torch.Generator().manual_seed(20 + seed)
            x = 3.0 * (refs[1:2].clone() + 0.02 * torch.randn(1, c, hw, hw, generator=g))   # un-normalised query
            proc = make_proc("fast_sdv3", "kernel_fast", refs, td, sigma=3.15, scale=0.03)
            xin = x.clone()
            out = proc.conditioning(xin)
            put(f"G4_s{seed}", x=x, refs=refs, scale=0.03, epsilon=1e-8, out_x=out["x_0_hat"])
        x = torch.randn(1, 16, 4, 4, generator=torch.Generator().manual_seed(5))
        x[:, :, 0, 0] = 0.0                                      # zero-norm pixel -> NaN propagation
        refs = chan_norm_refs(9, 16, 4, 300)
        proc = make_proc("fast_sdv3", "kernel_fast", refs, td, scale=0.03)
        out = proc.conditioning(x.clone())
        put("G4_nan", x=x, refs=refs, scale=0.03, epsilon=1e-8, out_x=out["x_0_hat"])

        # ---------------- G5: sparse ----------------
        for flavour in ("threshold", "fast", "fast_sdv3"):
            m, c, hw = 12, 4, 4
            refs = chan_norm_refs(m, c, hw, 400)
            g = torch.Generator().manual_seed(30)
            base = refs[3:4].clone() + 0.3 * torch.randn(1, c, hw, hw, generator=g)
            for tag, radius in (("some", 4.6), ("none", 0.05), ("all", 50.0)):
                proc = make_proc(flavour, "sparse", refs, td, radius=radius, scale=1.6)
                xin = base.clone()
                out = proc.conditioning(xin, beta_threshold=True) if flavour == "threshold" else proc.conditioning(xin)
                put(f"G5_{flavour}_{tag}", x=base, refs=refs, radius=radius, scale=1.6, out_x=out["x_0_hat"],
                    isneg=int(out.get("is_negation", -1)), force_norm=out["mean_x_0_hat"])

        # ---------------- G6: empirical_beta / empirical_radius ----------------
        m, c, hw = 10, 4, 4
        refs = chan_norm_refs(m, c, hw, 500)
        g = torch.Generator().manual_seed(40)
        noisy = {981: refs * 0.2 + 0.98 * torch.randn(refs.shape, generator=g),
                 1: refs * 0.999 + 0.03 * torch.randn(refs.shape, generator=g)}
        for q in (0.0, 0.25, 1.0):
            proc = make_proc("threshold", "kernel_fast", refs, td, sigma=3.15, beta_threshold=1.0, quantile=q)
            proc.noisy_proj_refs = noisy
            res = proc.empirical_beta(sigma=3.15, quantitle=q)
            put(f"G6_beta_q{q}", refs=refs, noisy981=noisy[981], noisy1=noisy[1], sigma=3.15, epsilon=1e-8, q=q,
                beta981=res[981], beta1=res[1])
            procs = make_proc("threshold", "sparse", refs, td, radius=1.0, quantile=q)
            procs.noisy_proj_refs = noisy
            res = procs.empirical_radius(quantitle=q)
            put(f"G6_radius_q{q}", refs=refs, noisy981=noisy[981], noisy1=noisy[1], q=q,
                radius981=res[981], radius1=res[1])

        # ---------------- G7: project() chunking + channel norm ----------------
        def fake_embed(img):                                   # 8x8 average pool, keep 3 channels + 1 derived
            z = torch.nn.functional.avg_pool2d(img, 8)
            return torch.cat([z, z.sum(1, keepdim=True)], 1)
        for flavour in ("threshold", "fast"):
            refs = chan_norm_refs(3, 4, 4, 600)
            proc = make_proc(flavour, "kernel_fast", refs, td, beta_threshold=1.0)
            proc.embed_fn = fake_embed
            for n in (3, 4, 9):                                # <, ==, > n_embed(=4)
                imgs = torch.randn(n, 3, 32, 32, generator=torch.Generator().manual_seed(n))
                put(f"G7_{flavour}_n{n}", imgs=imgs, n_embed=4, out=proc.project(imgs))

        # ---------------- G8: epsilon-dominated regime + full-size case ----------------
        m, c, hw = 515, 4, 64
        refs = chan_norm_refs(m, c, hw, 0)
        x = torch.randn(1, c, hw, hw, generator=torch.Generator().manual_seed(1000))
        proc = make_proc("fast", "kernel_fast", refs, td, scale=0.33)
        out = proc.conditioning(x.clone())
        ox = out["x_0_hat"].reshape(-1)
        put("G8_fast_full", seed_refs=0, seed_x=1000, m=m, scale=0.33, epsilon=1e-8,
            head=ox[:16], tail=ox[-16:], sum64=float(ox.double().sum()), l2_64=float(ox.double().norm()),
            max_abs_delta=float((out["x_0_hat"] - x).abs().max()))
        proc = make_proc("threshold", "kernel_fast", refs, td, sigma=3.15, scale=0.33, beta_threshold=2.0,
                         beta_threshold_margin=1.6)
        out = proc.conditioning(x.clone(), beta_threshold=True)
        ox = out["x_0_hat"].reshape(-1)
        put("G8_threshold_full", seed_refs=0, seed_x=1000, m=m, sigma=3.15, scale=0.33, epsilon=1e-8,
            beta_threshold=2.0, margin=1.6, head=ox[:16], tail=ox[-16:], sum64=float(ox.double().sum()),
            l2_64=float(ox.double().norm()), den=out["mean_x_0_hat"]["denominator"], isneg=int(out["is_negation"]))

    store["__cases__"] = np.array(meta)
    np.savez_compressed(OUT, **store)
    print(f"wrote {OUT}: {len(meta)} cases, {os.path.getsize(OUT)/1024:.1f} KiB")


if __name__ == "__main__":
    if not os.path.isdir("/root/reference/repellency"):
        sys.exit("reference not present: golden vectors can only be regenerated in the build container")
    if "--sdv3-full" in sys.argv:
        main_sdv3_full()
    else:
        main()
