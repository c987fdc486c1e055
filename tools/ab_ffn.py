#!/usr/bin/env python3
"""Fused GEGLU feed-forward (sdn_ffn_geglu_fused) against the two launches it replaces, in one process, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.unet import _interleave16  # noqa: E402
from tests_support import ops  # noqa: E402

B = int(os.environ.get("B", "128"))
C = 320
M = B * 4096
g = torch.Generator(device="cuda").manual_seed(0)
t = lambda *s, scale=1.0: (torch.randn(*s, generator=g, device="cuda") * scale).bfloat16()
x, res = t(M, C), t(M, C)
w1 = _interleave16(t(8 * C, C, scale=C ** -0.5)).contiguous(); b1 = torch.randn(8 * C, device="cuda")
gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
wcat, bcat = t(C, 5 * C, scale=(5 * C) ** -0.5), torch.randn(C, device="cuda")
cs = torch.zeros((M + 127) // 128, C, 2, device="cuda")


def two():
    ff = ops.gemm_ln(x, w1, gamma, beta, b1, act=2, prepass=True)
    return ops.gemm(ff, wcat, a2=x, bias=bcat, residual=res, col_stats=cs)


def one():
    return ops.ffn_fused(x, w1, gamma, beta, b1, wcat, bcat, res, col_stats=cs)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[2]


fl = 2.0 * M * (8 * C * C + C * 5 * C)
for rnd in range(2):
    a, b = timed(two), timed(one)
    print(f"B={B}: two launches (+fold, +row stats) {a:8.1f} us ({fl / a / 1e6:6.0f} TF)   fused (+fold, +row stats) {b:8.1f} us ({fl / b / 1e6:6.0f} TF)", flush=True)
