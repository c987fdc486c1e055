#!/usr/bin/env python3
"""In-kernel s_memtime stamps of the GEMM: per workgroup [start, first k-tile landed, k-loop done, epilogue done]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_denoiser_amd as sda  # noqa: E402
from tests_support import ops  # noqa: E402

B = int(os.environ.get("B", "32"))
import ctypes
lib = sda.lib()
lib.sdn_debug_set_gemm_stamps.argtypes = [ctypes.c_void_p]          # undeclared debug hook: 64-bit pointer, not int
lib.sdn_debug_set_gemm_stamps.restype = None
for name, M, N, K, conv, res in [("proj 320", B * 4096, 320, 320, None, True), ("proj 320 nores", B * 4096, 320, 320, None, False),
                                 ("ff2 320", B * 4096, 320, 1280, None, True), ("conv 320", B * 4096, 320, 2880, (64, 320), False)]:
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    if conv:
        H, Cin = conv
        a = torch.randn(M // (H * H), H, H, Cin, device="cuda").bfloat16(); kw = dict(conv=dict(Hs=H, Ws=H, Cin=Cin, Ho=H, Wo=H))
    else:
        a = torch.randn(M, K, device="cuda").bfloat16(); kw = {}
    r = torch.randn(M, N, device="cuda").bfloat16() if res else None
    ops.gemm(a, w, bias=bias, residual=r, **kw)
    st = torch.zeros(4 * 8192, dtype=torch.int64, device="cuda")
    lib.sdn_debug_set_gemm_stamps(st.data_ptr())
    ops.gemm(a, w, bias=bias, residual=r, **kw)
    torch.cuda.synchronize()
    lib.sdn_debug_set_gemm_stamps(None)
    s = st.cpu().reshape(-1, 4)
    s = s[s[:, 3] > 0]
    d = (s[:, 1:] - s[:, :-1]).double()
    tot = (s[:, 3] - s[:, 0]).double()
    span = float(s[:, 3].max() - s[:, 0].min())
    print(f"{name:16s} blocks {len(s):5d}  prologue+first DMA {d[:,0].median():8.0f}  k-loop {d[:,1].median():8.0f}  epilogue {d[:,2].median():8.0f}"
          f"  block total {tot.median():8.0f}   kernel span {span:9.0f}  (s_memtime ticks = shader cycles)")
