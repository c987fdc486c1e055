#!/usr/bin/env python3
"""SUSTAINED A/B of a GEMM debug variant on the UNet forward at B = 128: N forwards back to back per arm (N = 150 is ~16 s), arms
alternating.  Short interleaved rounds (tools/ab_gemm_variant.py) let a power-limited chip average its clock over both arms; this
form lets each arm settle.  VARIANT=128: next k-tile's DMA at the top of the iteration; 13: slab-ring convolution off.  To compare two
BUILDS on one box run it twice in one gpurun call, the second time with SDN_LIB=<other libsdn.so>."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import safe_denoiser_amd as sda
from safe_denoiser_amd.unet import UNet2DConditionModel
B = int(os.environ.get("B", "128"))
REP = int(os.environ.get("REP", "2"))
TEXTVER = os.environ.get("TEXTVER") == "1"       # second arm = the same library with the text version declared (K / V of the text reused)
u = UNet2DConditionModel(latent_repeat=REP)
u.load_synthetic_on_device(1234)
x = torch.randn(B // REP, 4, 64, 64, device="cuda")
e = u.prepare_text(torch.randn(B, 77, 768, device="cuda"))
y = torch.empty(B, 4, 64, 64, device="cuda")
N = int(os.environ.get("N", "150"))
for rnd in range(3):
    for v in (0, int(os.environ.get("VARIANT", "128"))):
        if os.environ.get("RESPRE") == "1":                      # second arm = attention output projections with the epilogue residual (res_pre off)
            import ctypes as C
            sda.lib().sdn_debug_set_res_pre(C.c_void_p(u._h.value), 0 if v else 1)
        elif os.environ.get("FFNSTATS") == "1":                  # second arm = norm3's row statistics by the sdn_row_stats pre-pass
            import ctypes as C
            sda.lib().sdn_debug_set_ffn_own_stats(C.c_void_p(u._h.value), 0 if v else 1)
        elif os.environ.get("LNPRE") == "1":                     # second arm = every folded LayerNorm's row statistics from the pre-pass (LNF = 2 kernels)
            import ctypes as C
            sda.lib().sdn_debug_set_ln_prepass_all(C.c_void_p(u._h.value), 1 if v else 0)
        elif TEXTVER:
            u.set_text_version(5 if v else 0)
        else:
            sda.lib().sdn_debug_set_gemm_variant(v)
        u._ws = {}
        u.forward_into(x, 981.0, e, y); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N): u.forward_into(x, 981.0, e, y)
        torch.cuda.synchronize()
        print(f"variant {v:3d}: {N} forwards back to back, {(time.perf_counter() - t0) / N * 1e3:7.2f} ms each", flush=True)
