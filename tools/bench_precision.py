"""Forward time of the SD-v1.4 UNet plan in each storage / contraction mode at one batch size (same-process comparison).
    python tools/bench_precision.py [B=32]
Modes: bf16, f16 (16-bit storage), bf16x3 (fp32 storage, split-operand contractions), fp32 (f32-input MFMA)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.unet import UNet2DConditionModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bf16", "bf16x3", "fp32"]
dev = torch.device("cuda")
for mode in modes:
    kw = dict(dtype=torch.bfloat16) if mode == "bf16" else dict(dtype=torch.float16) if mode == "f16" else dict(precision=mode)
    u = UNet2DConditionModel(latent_repeat=2, **kw)
    u.load_synthetic_on_device(1234, device=dev)
    x = torch.randn(B // 2, 4, 64, 64, device=dev)
    tb = u.prepare_text(torch.randn(B, 77, 768, device=dev))
    y = torch.empty((B, 4, 64, 64), device=dev)
    u.forward_into(x, 981.0, tb, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 3 if mode in ("fp32", "bf16x3") else 10
    e0.record()
    for _ in range(n):
        u.forward_into(x, 981.0, tb, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl, _ = u.flops(B)
    u.profile_next()
    u.forward_into(x, 981.0, tb, y)
    rows = sorted(u.profile_read(), key=lambda r: -r["ms"])
    print(f"{mode:7s} B={B}: {ms:8.2f} ms per forward, {fl / ms / 1e9:7.1f} TFLOP/s (algorithmic); finite={bool(torch.isfinite(y).all())}", flush=True)
    for r in rows[:6]:
        print(f"        {r['kernel']:20s} {r['launches']:4d} launches {r['ms']:8.2f} ms  {r['flops'] / max(r['ms'], 1e-9) / 1e9:7.1f} TFLOP/s", flush=True)
    del u, x, tb, y
    torch.cuda.empty_cache()
