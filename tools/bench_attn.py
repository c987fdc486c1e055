#!/usr/bin/env python3
"""Attention kernel timing on the SD-v1.4 shapes (median of 7 rounds)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests_support import ops  # noqa: E402

import safe_denoiser_amd as sda  # noqa: E402

B = int(os.environ.get("B", "32"))
if os.environ.get("QS2") is not None:
    sda.lib().sdn_debug_set_attn_qs2(int(os.environ["QS2"]))
if os.environ.get("NOMAX") is not None:
    sda.lib().sdn_debug_set_attn_nomax(int(os.environ["NOMAX"]))
if os.environ.get("HEAD_INNER") is not None:
    sda.lib().sdn_debug_set_attn_head_inner(int(os.environ["HEAD_INNER"]))
for name, nq, nk, d in [("self 64x64 d40", 4096, 4096, 40), ("self 32x32 d80", 1024, 1024, 80),
                        ("self 16x16 d160", 256, 256, 160), ("self 8x8 d160", 64, 64, 160),
                        ("cross 64x64 d40", 4096, 77, 40), ("cross 32x32 d80", 1024, 77, 80)]:
    q = torch.randn(B, nq, 8 * d, device="cuda").bfloat16()
    k = torch.randn(B, nk, 8 * d, device="cuda").bfloat16()
    v = torch.randn(B, nk, 8 * d, device="cuda").bfloat16()
    ops.attention(q, k, v, 8)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.attention(q, k, v, 8)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3 * 1e3)
    t = sorted(ts)[3]
    fl = 4.0 * B * 8 * nq * nk * d
    print(f"{name:20s} {t:9.1f} us  {fl / t / 1e6:7.1f} TF/s")
