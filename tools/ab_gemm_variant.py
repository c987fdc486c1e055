#!/usr/bin/env python3
"""In-process A/B of a GEMM debug variant (sdn_debug_set_gemm_variant) on the whole UNet forward at B = 128, interleaved rounds.
VARIANT=128: the next k-tile's DMA at the top of the iteration (round-1 placement); 9: round-1 tile rule; 3: no 256-row tile."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_denoiser_amd as sda  # noqa: E402
from safe_denoiser_amd.unet import UNet2DConditionModel  # noqa: E402

B = int(os.environ.get("B", "128"))
V = int(os.environ.get("VARIANT", "128"))
u = UNet2DConditionModel(latent_repeat=2)
u.load_synthetic_on_device(1234)
x = torch.randn(B // 2, 4, 64, 64, device="cuda")
e = u.prepare_text(torch.randn(B, 77, 768, device="cuda"))
y = torch.empty(B, 4, 64, 64, device="cuda")
outs = {}
res = {0: [], V: []}
for rnd in range(4):
    for v in (0, V):
        sda.lib().sdn_debug_set_gemm_variant(v)
        u._ws = {}
        u.forward_into(x, 981.0, e, y)
        torch.cuda.synchronize()
        outs[v] = y.clone()
        t0 = time.perf_counter()
        for _ in range(5):
            u.forward_into(x, 981.0, e, y)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 5 * 1e3)
sda.lib().sdn_debug_set_gemm_variant(0)
for v in (0, V):
    r = sorted(res[v])
    print(f"variant {v:3d}: forward ms per round {['%.2f' % t for t in res[v]]}  median {r[len(r) // 2]:.2f}")
print("outputs bit-identical:", bool(torch.equal(outs[0], outs[V])))
