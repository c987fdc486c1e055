#!/usr/bin/env python3
"""Capture golden vectors for the SAFREE text-projection helpers from the reference's own source.

Runs ONLY in the build container (reads /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_safree_golden.py

The module models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py imports diffusers at file
scope and cannot be imported here, but its helpers (`sigmoid`, `f_beta`, `projection_matrix`,
`projection_and_orthogonal`, `safree_projection`, lines 16-99) are pure torch/math top-level functions: they are
picked out of the parsed module by name and executed as they stand.  Only INPUT and OUTPUT arrays are written
(tests/golden/safree_golden.npz); no reference source text travels.
"""
import ast
import contextlib
import io
import math
import os

import numpy as np
import torch

SRC = "/root/reference/models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py"
SRC_SD3 = "/root/reference/models/sdv3/safe_denoiser_pipeline.py"      # projection_matrix / mask_to_onp, lines 71-153
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "safree_golden.npz")
WANT = ("sigmoid", "f_beta", "projection_matrix", "projection_and_orthogonal", "safree_projection")


def load_helpers(src=SRC, want=WANT):
    tree = ast.parse(open(src).read(), src)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert sorted(n.name for n in body) == sorted(want), [n.name for n in body]
    ns = {"torch": torch, "math": math, "F": torch.nn.functional}
    exec(compile(ast.Module(body=body, type_ignores=[]), src, "exec"), ns)
    return ns


def main():
    ref = load_helpers()
    store = {}
    torch.set_num_threads(4)

    # ---- f_beta: grid of z x {sigmoid, tanh} x {nudity, artists-*} x upper bounds ----
    zs = np.linspace(-0.2, 1.2, 57)
    rows = []
    for btype in ("sigmoid", "tanh"):
        for concept in ("nudity", "artists-VanGogh"):
            for up_t in (10, 20):
                rows.append([ref["f_beta"](float(z), btype=btype, upperbound_timestep=up_t, concept_type=concept) for z in zs])
    store["f_beta/z"] = zs
    store["f_beta/out"] = np.asarray(rows, dtype=np.int64)         # order: btype outer, concept, up_t inner

    # ---- projection_matrix / safree_projection / projection_and_orthogonal ----
    # cases: (dim, n_tokens, n_negative_phrases, alpha, dtype, plant a trigger token?)
    cases = [(96, 9, 17, 0.01, torch.float64, True), (96, 5, 17, 0.0, torch.float64, False),
             (64, 1, 3, 0.01, torch.float64, False),               # single token: leave-one-out mean of nothing = nan
             (768, 12, 17, 0.01, torch.float32, True), (256, 75, 17, 0.01, torch.float32, True)]
    names = []
    for ci, (dim, n_t, n_neg, alpha, dt, plant) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + ci)
        ie = torch.randn(2, 77, dim, generator=g, dtype=dt)
        neg = torch.randn(n_neg, dim, generator=g, dtype=dt)
        p_emb = torch.randn(n_t, dim, generator=g, dtype=dt)
        if plant:
            p_emb[min(3, n_t - 1)] = neg[:4].mean(0) * 3 + 0.05 * p_emb[min(3, n_t - 1)]
        P_c = ref["projection_matrix"](neg.T)
        P_m = ref["projection_matrix"](p_emb.T)
        with contextlib.redirect_stdout(io.StringIO()) as log:
            resc = ref["safree_projection"](ie, p_emb, P_m, P_c, alpha=alpha, max_length=77, logger=None)
        n_removed = int(log.getvalue().split("we remove")[1].strip().rstrip("."))
        ort = ref["projection_and_orthogonal"](ie, P_m, P_c)
        name = f"case{ci}"
        names.append(name)
        arrs = dict(ie=ie, neg=neg, p_emb=p_emb, rescaled=resc, proj_ort=ort)
        if dim <= 96:                                                # the projectors themselves only for the small cases
            arrs.update(P_c=P_c, P_m=P_m)                            # (fixture size; the large cases pin them through their use)
        for k, v in arrs.items():
            store[f"{name}/{k}"] = v.numpy()
        store[f"{name}/alpha"] = np.float64(alpha)
        store[f"{name}/n_removed"] = np.int64(n_removed)
    # ---- SD-v3 variant: projection_matrix (fp32 pinverse) + mask_to_onp (bf16 products), a 333-token axis ----
    sd3 = load_helpers(SRC_SD3, ("projection_matrix", "mask_to_onp"))
    sd3_names = []
    for ci, (dim, n_t, n_neg, alpha, dt) in enumerate([(128, 9, 17, 0.01, torch.float32), (256, 14, 17, 0.01, torch.float16)]):
        g = torch.Generator().manual_seed(300 + ci)
        ie = torch.randn(2, 333, dim, generator=g).to(dt)
        neg = torch.randn(n_neg, dim, generator=g).to(dt)
        p_emb = torch.randn(n_t, dim, generator=g).to(dt)
        p_emb[3] = (neg[:4].float().mean(0) * 3 + 0.05 * p_emb[3].float()).to(dt)
        P_m = sd3["projection_matrix"](p_emb.T)
        P_c = sd3["projection_matrix"](neg.T)
        with contextlib.redirect_stdout(io.StringIO()):
            resc, keep, inv, n_removed = sd3["mask_to_onp"](ie, p_emb, P_m, P_c, alpha=alpha, debug=False)
        name = f"sd3_case{ci}"
        sd3_names.append(name)
        for k, v in dict(ie=ie, neg=neg, p_emb=p_emb, rescaled=resc, keep=keep, inv=inv).items():
            store[f"{name}/{k}"] = v.float().numpy()
        store[f"{name}/alpha"] = np.float64(alpha)
        store[f"{name}/n_removed"] = np.float64(n_removed)
        store[f"{name}/f16"] = np.int64(dt == torch.float16)
    store["__sd3_cases__"] = np.asarray(sd3_names)
    store["__cases__"] = np.asarray(names)
    np.savez_compressed(OUT, **store)
    print(f"wrote {OUT}: f_beta grid {store['f_beta/out'].shape}, {len(names)} projection cases")


if __name__ == "__main__":
    main()
