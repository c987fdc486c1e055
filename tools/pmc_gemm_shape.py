#!/usr/bin/env python3
"""One projection shape run a few times, for `rocprofv3 --pmc ... -- python3 tools/pmc_gemm_shape.py` (SHAPE=ff1|qkv|proj|conv)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.unet import _interleave16  # noqa: E402
from tests_support import ops  # noqa: E402

B = int(os.environ.get("B", "128"))
shape = os.environ.get("SHAPE", "ff1")
g = torch.Generator(device="cuda").manual_seed(0)
t = lambda *s, scale=1.0: (torch.randn(*s, generator=g, device="cuda") * scale).bfloat16()
M = B * 4096
if shape == "ff1":
    a, w, bias = t(M, 320), _interleave16(t(2560, 320, scale=320 ** -0.5)).contiguous(), torch.randn(2560, device="cuda")
    gamma, beta = torch.ones(320, device="cuda"), torch.zeros(320, device="cuda")
    fn = lambda: ops.gemm_ln(a, w, gamma, beta, bias, act=2, prepass=True)
elif shape == "qkv":
    a, w, bias = t(M, 320), t(960, 320, scale=320 ** -0.5), torch.randn(960, device="cuda")
    gamma, beta = torch.ones(320, device="cuda"), torch.zeros(320, device="cuda")
    fn = lambda: ops.gemm_ln(a, w, gamma, beta, bias, prepass=True)
elif shape == "proj":
    a, w, bias, r = t(M, 320), t(320, 320, scale=320 ** -0.5), torch.randn(320, device="cuda"), t(M, 320)
    fn = lambda: ops.gemm(a, w, bias=bias, residual=r)
else:
    a, w, bias = t(B, 64, 64, 320), t(320, 2880, scale=2880 ** -0.5), torch.randn(320, device="cuda")
    fn = lambda: ops.gemm(a, w, bias=bias, conv=dict(Hs=64, Ws=64, Cin=320, Ho=64, Wo=64))
for _ in range(4):
    fn()
torch.cuda.synchronize()
