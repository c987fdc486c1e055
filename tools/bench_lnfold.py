#!/usr/bin/env python3
"""LayerNorm + GEMM (two kernels, LN output through HBM) vs the LayerNorm-folded GEMM (sdn_gemm_ln_*)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.unet import _interleave16  # noqa: E402
from tests_support import ops  # noqa: E402

B = int(os.environ.get("B", "64"))


def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, N, K, geglu in [("qkv 320", B * 4096, 960, 320, False), ("to_q 320", B * 4096, 320, 320, False), ("ff1 320", B * 4096, 2560, 320, True),
                             ("qkv 640", B * 1024, 1920, 640, False), ("ff1 640", B * 1024, 5120, 640, True), ("ff1 1280", B * 256, 10240, 1280, True)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    g, b_, bias = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda") * 0.1, torch.randn(N, device="cuda")
    if geglu:
        w, bias = _interleave16(w).contiguous(), _interleave16(bias).contiguous()
    act = 2 if geglu else 0
    t_ln = timeit(lambda: ops.layernorm(x, g, b_))
    ln = ops.layernorm(x, g, b_)
    t_mm = timeit(lambda: ops.gemm(ln, w, bias=bias, act=act))
    t_f = timeit(lambda: ops.gemm_ln(x, w, g, b_, bias, act=act)) if (N % 160 == 0 and not geglu) or N % 64 == 0 and N < 128 else float("nan")
    t_p = timeit(lambda: ops.gemm_ln(x, w, g, b_, bias, act=act, prepass=True))       # both include the (tiny) fold kernel
    print(f"{name:10s} LN {t_ln:7.1f} + GEMM {t_mm:7.1f} = {t_ln + t_mm:7.1f} us   folded, in-kernel stats {t_f:7.1f} us   "
          f"pre-pass stats {t_p:7.1f} us ({100 * (1 - t_p / (t_ln + t_mm)):+.1f} %)")
