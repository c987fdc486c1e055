#!/usr/bin/env python3
"""VAE decoder timing at the SD-v1.4 size (64x64x4 latents -> 512x512 images): ms per image, TFLOP/s, per-kernel rows."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.vae import AutoencoderKL  # noqa: E402

B = int(os.environ.get("B", "8"))
v = AutoencoderKL()
v.load_synthetic_on_device(5)
z = torch.randn(B, 4, 64, 64, device="cuda") * 0.18215
v.decode_latents_uint8(z)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    v.decode_latents_uint8(z)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = sorted(ts)[2]
fl, _ = v.flops(min(B, v.MAX_CHUNK))
fl *= B / min(B, v.MAX_CHUNK)
print(f"B={B}: {t:.2f} ms per batch, {t / B:.2f} ms per image, {fl / t / 1e9:.1f} TFLOP/s ({fl / B / 1e12:.3f} TFLOP per image)")
v.profile_next()
v.decode(z[:min(B, v.MAX_CHUNK)])
rows = sorted(v.profile_read(), key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in rows)
for r in rows:
    tf = r["flops"] / r["ms"] / 1e9 if r["flops"] else 0.0
    print(f"  {r['kernel']:18s} x{r['launches']:4d} {r['ms']:8.3f} ms {100 * r['ms'] / tot:5.1f} %  {tf:7.1f} TF/s  {r['bytes'] / r['ms'] / 1e6:7.1f} GB/s")
