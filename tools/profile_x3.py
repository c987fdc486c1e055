"""Per-label profile of one fp32-storage UNet forward (HIP events around every launch of the plan): where the bf16x3 mode's
time goes.  PREC=bf16x3|fp32, BATCH (samples = guidance branches x prompts), REP (latent_repeat)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.unet import UNet2DConditionModel  # noqa: E402

prec = os.environ.get("PREC", "bf16x3")
B = int(os.environ.get("BATCH", "96"))
rep = int(os.environ.get("REP", "3"))
u = UNet2DConditionModel(latent_repeat=rep, precision=prec)
u.load_synthetic_on_device(1234)
if os.environ.get("PAIRS") == "0":                 # A/B: f32 qkv tensor + sdn_attention_x3 instead of the pre-split self-attention
    import ctypes as C
    import safe_denoiser_amd as sda
    sda.lib().sdn_debug_set_x3_pairs(C.c_void_p(u._h.value), 0)
x = torch.randn(B // rep, 4, 64, 64, device="cuda")
tb = u.prepare_text(torch.randn(B, 77, 768, device="cuda"))
y = torch.empty(B, 4, 64, 64, device="cuda")
for _ in range(2):
    u.forward_into(x, 981.0, tb, y)
torch.cuda.synchronize()
acc = {}
for _ in range(2):
    u.profile_next()
    u.forward_into(x, 981.0, tb, y)
    for r in u.profile_read():
        a = acc.setdefault(r["kernel"], dict(launches=0, ms=0.0, flops=0.0))
        a["launches"] += r["launches"]; a["ms"] += r["ms"]; a["flops"] += r["flops"]
tot = sum(v["ms"] for v in acc.values()) / 2
fl, _ = u.flops(B)
print(f"{prec} forward at B = {B} (latent_repeat {rep}): {tot:.1f} ms = {fl / tot / 1e9:.0f} TFLOP/s algorithmic")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:22s} {v['launches'] // 2:4d} launches {v['ms'] / 2:8.2f} ms {100 * v['ms'] / 2 / tot:5.1f} %  "
          f"{(v['flops'] / v['ms'] / 1e9) if v['flops'] else 0:7.0f} TFLOP/s")
