#!/usr/bin/env python3
"""Phase timing of the attention main loop from in-kernel s_memtime stamps.  Needs the diagnostics build:
   make -C safe_denoiser_amd/csrc stamps   (-> safe_denoiser_amd/libsdn_stamps.so, -DSDN_ATTN_STAMPS)
   SDN_LIB=$PWD/safe_denoiser_amd/libsdn_stamps.so python tools/attn_stamps.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_denoiser_amd as sda  # noqa: E402
from tests_support import ops  # noqa: E402

B = int(os.environ.get("B", "32"))
lib = sda.lib()
lib.sdn_debug_set_attn_stamps.argtypes = [ctypes.c_void_p]
names = ["issue DMA + K reads + QK MFMAs", "max / exp2 (waits for S)", "pack + V tr reads + PV MFMAs", "vmcnt(0) next tile", "barrier"]
B = int(os.environ.get("B", "32"))
for name, nq, nk, d in [("self 64x64 d40", 4096, 4096, 40), ("self 32x32 d80", 1024, 1024, 80), ("cross 64x64 d40", 4096, 77, 40),
                        ("self 16x16 d160", 256, 256, 160)]:
    q = torch.randn(B, nq, 8 * d, device="cuda").bfloat16()
    k = torch.randn(B, nk, 8 * d, device="cuda").bfloat16()
    v = torch.randn(B, nk, 8 * d, device="cuda").bfloat16()
    nblk = B * 8 * ((nq + 127) // 128)
    st = torch.zeros(nblk * 4 * 7, dtype=torch.int64, device="cuda")
    ops.attention(q, k, v, 8)
    lib.sdn_debug_set_attn_stamps(st.data_ptr())
    ops.attention(q, k, v, 8)
    torch.cuda.synchronize()
    lib.sdn_debug_set_attn_stamps(None)
    s7 = st.cpu().reshape(-1, 7).double()
    s7 = s7[s7[:, :5].sum(1) > 0]
    s = s7[:, :5]
    it = (nk + 63) // 64
    tot = s.sum(1).median() / it
    print(f"{name}: {len(s)} waves, {it} iterations, {tot:7.0f} ticks per wave-iteration")
    for i, n in enumerate(names):
        print(f"    {n:34s} {s[:, i].median() / it:8.0f}  ({100 * s[:, i].median() / s.sum(1).median():4.1f} %)")
    tot_all = (s7[:, 5] + s7[:, 6] + s.sum(1)).median()
    print(f"    per wave: prologue {s7[:, 5].median():8.0f}  main loop {s.sum(1).median():9.0f}  epilogue {s7[:, 6].median():7.0f}"
          f"   -> loop share {100 * s.sum(1).median() / tot_all:4.1f} %")
