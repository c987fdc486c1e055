#!/usr/bin/env python3
"""In-process A/B of the FeedForward-output + proj_out contraction (sdn_debug_set_ff_fuse): forward time at B = 128 with the
fusion on / off, interleaved rounds (guide rule 24), plus the per-shape profile of the fused GEMMs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_denoiser_amd as sda  # noqa: E402
from safe_denoiser_amd.unet import UNet2DConditionModel  # noqa: E402

B = int(os.environ.get("B", "128"))
u = UNet2DConditionModel(latent_repeat=2)
u.load_synthetic_on_device(1234)
x = torch.randn(B // 2, 4, 64, 64, device="cuda")
e = u.prepare_text(torch.randn(B, 77, 768, device="cuda"))
y = torch.empty(B, 4, 64, 64, device="cuda")
res = {0: [], 1: []}
for rnd in range(4):
    for on in (1, 0):
        getattr(sda.lib(), os.environ.get("HOOK", "sdn_debug_set_ff_fuse"))(u._h, on)
        u._ws = {}
        u.forward_into(x, 981.0, e, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            u.forward_into(x, 981.0, e, y)
        torch.cuda.synchronize()
        res[on].append((time.perf_counter() - t0) / 5 * 1e3)
for on in (1, 0):
    v = sorted(res[on])
    print(f"{os.environ.get('HOOK', 'sdn_debug_set_ff_fuse')}={on}: forward ms per round {['%.2f' % t for t in res[on]]}  median {v[len(v) // 2]:.2f}  min {v[0]:.2f}")
