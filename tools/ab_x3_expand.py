"""Same-process A/B of the two bf16x3 GEMM paths at one batch: f32-staging k_gemm_x3 (expand off) vs triple operands on the LDS-DMA
tiles (expand on), plus the distance between their outputs.  BATCH / REP as tools/profile_x3.py."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_denoiser_amd as sda  # noqa: E402
from safe_denoiser_amd.unet import UNet2DConditionModel  # noqa: E402

B = int(os.environ.get("BATCH", "96"))
rep = int(os.environ.get("REP", "3"))
u = UNet2DConditionModel(latent_repeat=rep, precision="bf16x3")
u.load_synthetic_on_device(1234)
x = torch.randn(B // rep, 4, 64, 64, device="cuda")
tb = u.prepare_text(torch.randn(B, 77, 768, device="cuda"))
outs = {}
for on in (1, 0, 1):
    sda.lib().sdn_debug_set_x3_expand(C.c_void_p(u._h.value), on)
    u._ws = {}
    y = torch.empty(B, 4, 64, 64, device="cuda")
    for _ in range(2):
        u.forward_into(x, 981.0, tb, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        u.forward_into(x, 981.0, tb, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    fl, _ = u.flops(B)
    print(f"x3 expand {on}: {ms:.1f} ms per forward at B = {B} = {fl / ms / 1e9:.0f} TFLOP/s algorithmic", flush=True)
    outs[on] = y.double().clone()
print("rel L2 between the two paths:", float((outs[1] - outs[0]).norm() / outs[0].norm()))
