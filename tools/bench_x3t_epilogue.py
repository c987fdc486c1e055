#!/usr/bin/env python3
"""What the bf16x3 plan's epilogue forms cost on a narrow-N, HBM-streamed GEMM (N = 320, K = 1280 -> K' = 3840): plain 16-bit GEMM of the
same k loop vs x3_out = 1 (f32 rows) with / without the f32 residual vs x3_out = 3 (triple).  B samples of 4096 rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests_support import ops  # noqa: E402
B = int(os.environ.get("B", "48"))
for N, K in ((320, 1280), (320, 320), (640, 2560)):
    M = B * (4096 if N == 320 else 1024)
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda")
    a3, w3 = ops.split3(a), ops.expand3(w)
    a16, w16 = a3.contiguous(), w3.contiguous()                       # the same bytes as a plain bf16 GEMM with K' = 3K
    forms = [("bf16 GEMM, K' = 3K, 16-bit out", lambda: ops.gemm(a16, w16, bias=bias)),
             ("x3_out = 1 (f32 rows)", lambda: ops.gemm_x3t(a3, w3, N, K, bias=bias)),
             ("x3_out = 1 + f32 residual", lambda: ops.gemm_x3t(a3, w3, N, K, bias=bias, residual=res)),
             ("x3_out = 3 (triple)", lambda: ops.gemm_x3t(a3, w3, N, K, bias=bias, x3_out=3)),
             ("x3_out = 3 + f32 residual", lambda: ops.gemm_x3t(a3, w3, N, K, bias=bias, residual=res, x3_out=3))]
    for name, fn in forms:
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): fn()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 3 * 1e3)
        t = sorted(ts)[2]
        print(f"M={M} N={N} K={K}  {name:34s} {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s algorithmic", flush=True)
