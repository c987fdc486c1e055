#!/usr/bin/env python3
"""Repellency projection micro-benchmark: sdn_repel_apply at the SD-v1.4 shape (M = 515, D = 16384) for N = 1 / 8 / 64 queries
(and the SD-v3 shape with SD3=1).  Prints microseconds per call and algorithmic HBM GB/s (one read of proj_ref + x in/out).
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.repellency import repellency_methods_threshold as thr  # noqa: E402

M = int(os.environ.get("REFS", "515"))
C, S = (16, 64) if os.environ.get("SD3") else (4, 64)
refs = torch.randn(M, C, S, S, generator=torch.Generator().manual_seed(0))
refs = refs / refs.norm(dim=1, keepdim=True)
path = os.path.join(tempfile.mkdtemp(), "pr.pt")
torch.save(refs, path)
proc = thr.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085, 0.012, n_embed=16,
                                 proj_ref_path=path, cache_proj_ref=True, scale=0.33, sigma=3.15, beta_threshold=1.0,
                                 beta_threshold_margin=1.6)
for N in (1, 4, 8, 16, 64):
    x = torch.randn(N, C, S, S, device="cuda")
    for _ in range(3):
        proc.conditioning_device(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        proc.conditioning_device(x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    by = M * C * S * S * 4 + 2 * N * C * S * S * 4
    print(f"N={N:3d}: {us:8.1f} us per call, {by / us / 1e3:7.1f} GB/s algorithmic ({by / us / 1e3 / 8000 * 100:.1f} % of 8 TB/s), "
          f"{5.0 * N * M * C * S * S / us / 1e6:.1f} TFLOP/s")
