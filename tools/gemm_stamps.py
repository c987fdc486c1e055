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
                                 ("ff2 320", B * 4096, 320, 1280, None, True), ("conv 320", B * 4096, 320, 2880, (64, 320), False),
                                 ("qkv 320", B * 4096, 960, 320, None, False), ("ff1 geglu 320", B * 4096, 2560, 320, None, "geglu"),
                                 ("ff1 geglu 640", B * 1024, 5120, 640, None, "geglu")]:
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    if conv:
        H, Cin = conv
        a = torch.randn(M // (H * H), H, H, Cin, device="cuda").bfloat16(); kw = dict(conv=dict(Hs=H, Ws=H, Cin=Cin, Ho=H, Wo=H))
    else:
        a = torch.randn(M, K, device="cuda").bfloat16(); kw = {}
    if res == "geglu":
        kw["act"] = 2
        res = False
    r = torch.randn(M, N, device="cuda").bfloat16() if res else None
    ops.gemm(a, w, bias=bias, residual=r, **kw)
    st = torch.zeros(8 * 32768, dtype=torch.int64, device="cuda")
    lib.sdn_debug_set_gemm_stamps(st.data_ptr())
    ops.gemm(a, w, bias=bias, residual=r, **kw)
    torch.cuda.synchronize()
    lib.sdn_debug_set_gemm_stamps(None)
    s8 = st.cpu().reshape(-1, 8)
    s8 = s8[s8[:, 3] > 0]
    s = s8[:, :4]
    s = s[s[:, 3] > 0]
    d = (s[:, 1:] - s[:, :-1]).double()
    tot = (s[:, 3] - s[:, 0]).double()
    print(f"{name:16s} blocks {len(s):5d}  prologue+first DMA {d[:,0].median():8.0f}  k-loop {d[:,1].median():8.0f}  epilogue {d[:,2].median():8.0f}"
          f"  block total {tot.median():8.0f}")
    e = s8.double()
    print(f"{'':16s} prologue: index math {(e[:,7]-e[:,0]).median():7.0f}  first DMA issue+wait {(e[:,1]-e[:,7]).median():7.0f}")
    print(f"{'':16s} epilogue pass 0: acc->LDS {(e[:,4]-e[:,2]).median():7.0f}  barrier {(e[:,5]-e[:,4]).median():7.0f}  store loop {(e[:,6]-e[:,5]).median():7.0f}"
          f"  rest (pass 1 / tail) {(e[:,3]-e[:,6]).median():7.0f}   [s_memtime ticks]")
