"""Per-shape timing of sdn_gemm_x3 (fp32 storage, bf16x3 products) with the timing-only ablations of SDN_X3_DBG.
    SDN_X3_DBG=<bits> python tools/bench_x3_gemm.py        (1 no split arithmetic, 2 no loads in the loop, 4 no LDS writes, 8 one MFMA per product)
The ablation branches are compiled in only with -DSDN_X3_ABLATE (make -C safe_denoiser_amd/csrc CXXFLAGS+=-DSDN_X3_ABLATE after touching
sdn_f32.hip): as runtime branches they cost the production kernel 13 %."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests_support import ops  # noqa: E402

ops.X3 = True
B = int(os.environ.get("B", "32"))
shapes = [("conv 320->320 @64", dict(conv=dict(Hs=64, Ws=64, Cin=320, Ho=64, Wo=64)), (B, 64, 64, 320), (320, 2880)),
          ("conv 640->640 @32", dict(conv=dict(Hs=32, Ws=32, Cin=640, Ho=32, Wo=32)), (B, 32, 32, 640), (640, 5760)),
          ("FF1 640 (N=5120)", dict(), (B * 1024, 640), (5120, 640)),
          ("proj 320 (N=320,K=320)", dict(), (B * 4096, 320), (320, 320))]
for name, kw, ashape, wshape in shapes:
    a = torch.randn(*ashape, device="cuda")
    w = torch.randn(*wshape, device="cuda") * wshape[1] ** -0.5
    ops.gemm(a, w, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm(a, w, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    M = a.numel() // ashape[-1]
    fl = 2.0 * M * wshape[0] * wshape[1]
    print(f"dbg={os.environ.get('SDN_X3_DBG', '0'):>2s} {name:24s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s (algorithmic)", flush=True)
