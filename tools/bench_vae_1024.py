#!/usr/bin/env python3
"""SD-v3 VAE decode at 1024x1024 (16-channel latents 128x128 -> images): ms per image."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.vae import SD3_VAE_CONFIG, AutoencoderKL  # noqa: E402

v = AutoencoderKL(**dict(SD3_VAE_CONFIG, sample_size=1024))
v.load_synthetic_on_device(3)
n = v.MAX_CHUNK
z = torch.randn(n, 16, 128, 128, device="cuda")
u = v.decode_latents_uint8(z); torch.cuda.synchronize()
t0 = time.perf_counter(); u = v.decode_latents_uint8(z); torch.cuda.synchronize(); dt = time.perf_counter() - t0
fl, _ = v.flops(n)
print(f"SD3 VAE 1024x1024: {tuple(u.shape)} {dt / n * 1e3:.1f} ms per image ({fl / dt / 1e12:.0f} TFLOP/s), {n} images per plan invocation")
