#!/usr/bin/env python3
"""UNet-only workload for counter passes: N forwards of the SD-v1.4 plan at the benchmark's batch (3 branches x 64 prompts = 192
samples, latent_repeat 3, bf16) and nothing else -- no CLIP, VAE, repellency or torch elementwise kernels sharing a symbol with the
plan's (the round-3 traffic record mixed the VAE decoder's launches of k_gemm_dma into the UNet's).
    python3 tools/unet_forward.py [forwards=2] [batch=192] [dtype=bf16|f16|bf16x3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.unet import UNet2DConditionModel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 192
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16"
kw = dict(precision="bf16x3") if mode == "bf16x3" else dict(dtype=torch.float16 if mode == "f16" else torch.bfloat16)
u = UNet2DConditionModel(latent_repeat=3, **kw)
u.load_synthetic_on_device(1234)
x = torch.randn(B // 3, 4, 64, 64, device="cuda")
tb = u.prepare_text(torch.randn(B, 77, 768, device="cuda"))
y = torch.empty(B, 4, 64, 64, device="cuda")
for _ in range(n):
    u.forward_into(x, 981.0, tb, y)
torch.cuda.synchronize()
print("forwards done", n, B, mode, bool(torch.isfinite(y).all()))
