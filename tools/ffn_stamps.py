#!/usr/bin/env python3
"""Phase timing of the fused GEGLU feed-forward (csrc/sdn_ffn.hip) from in-kernel s_memtime stamps.  Needs the diagnostics build:
   make -C safe_denoiser_amd/csrc stamps
   SDN_LIB=$PWD/safe_denoiser_amd/libsdn_stamps.so python tools/ffn_stamps.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_denoiser_amd as sda  # noqa: E402
from safe_denoiser_amd.unet import _interleave16  # noqa: E402
from tests_support import ops  # noqa: E402

B = int(os.environ.get("B", "128"))
C = 320
M = B * 4096
lib = sda.lib()
lib.sdn_debug_set_ffn_stamps.argtypes = [ctypes.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
t = lambda *s, scale=1.0: (torch.randn(*s, generator=g, device="cuda") * scale).bfloat16()
x, res = t(M, C), t(M, C)
w1 = _interleave16(t(8 * C, C, scale=C ** -0.5)).contiguous(); b1 = torch.randn(8 * C, device="cuda")
gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
wcat, bcat = t(C, 5 * C, scale=(5 * C) ** -0.5), torch.randn(C, device="cuda")
nblk = (M + 127) // 128
st = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device="cuda")
ops.ffn_fused(x, w1, gamma, beta, b1, wcat, bcat, res)
lib.sdn_debug_set_ffn_stamps(st.data_ptr())
ops.ffn_fused(x, w1, gamma, beta, b1, wcat, bcat, res)
torch.cuda.synchronize()
lib.sdn_debug_set_ffn_stamps(None)
s = st.cpu().reshape(-1, 8).double()
s = s[s.sum(1) > 0]
names = ["prologue (X + first W1 k-tile)", "projection: issue, reads, MFMAs", "projection: wait for next k-tile", "projection: barrier",
         "GEGLU epilogue", "barriers around the contraction", "contraction: reads, MFMAs", "trailing k-tiles + final epilogue"]
tot = s.sum(1).median()
print(f"{len(s)} waves; {tot:9.0f} ticks per wave = one 128-row block (20 chunks)")
for i, n in enumerate(names):
    per = s[:, i].median()
    print(f"   {n:40s} {per:9.0f}  ({100 * per / tot:4.1f} %)" + (f"   {per / 20:7.0f} per chunk" if 1 <= i <= 6 else ""))
