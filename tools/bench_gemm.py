#!/usr/bin/env python3
"""In-process A/B of GEMM kernel variants on the SD-v1.4 UNet's shapes (interleaved rounds, median; guide rule 24)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_denoiser_amd as sda  # noqa: E402
from tests_support import ops  # noqa: E402

BF = torch.bfloat16
B = int(os.environ.get("B", "32"))
SHAPES = [  # name, M, N, K, conv(H, Cin) or None, act
    ("conv 320->320 @64", B * 4096, 320, 2880, (64, 320), 0),
    ("conv 960->320 @64", B * 4096, 320, 8640, (64, 960), 0),
    ("conv 640->640 @32", B * 1024, 640, 5760, (32, 640), 0),
    ("conv 1280->1280 @16", B * 256, 1280, 11520, (16, 1280), 0),
    ("conv 1280->1280 @8", B * 64, 1280, 11520, (8, 1280), 0),
    ("conv 2560->1280 @8", B * 64, 1280, 23040, (8, 2560), 0),
    ("qkv 320 @64", B * 4096, 960, 320, None, 0),
    ("proj 320 @64", B * 4096, 320, 320, None, 0),
    ("ff1 geglu 320 @64", B * 4096, 2560, 320, None, 2),
    ("ff2 320 @64", B * 4096, 320, 1280, None, 0),
    ("ff1 geglu 640 @32", B * 1024, 5120, 640, None, 2),
    ("ff2 640 @32", B * 1024, 640, 2560, None, 0),
    ("ff1 geglu 1280 @16", B * 256, 10240, 1280, None, 2),
    ("qkv 640 @32", B * 1024, 1920, 640, None, 0),
    ("qkv 1280 @16", B * 256, 3840, 1280, None, 0),
    ("x3k ff2 320", B * 4096, 320, 3840, None, 0),          # the bf16x3 plan's k loops (K' = 3 K) on narrow N: A streamed once from HBM
    ("x3k proj 320", B * 4096, 320, 960, None, 0),
    ("x3k ff2 640", B * 1024, 640, 7680, None, 0),
    ("x3k proj 640", B * 1024, 640, 1920, None, 0),
    ("x3k proj 1280", B * 256, 1280, 3840, None, 0),
    ("mmdit qkv 1536", B * 1024, 4608, 1536, None, 0),
    ("mmdit ff1 1536", B * 1024, 6144, 1536, None, 3),
]
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,1").split(",")]
if os.environ.get("SKIP"):                                   # comma-separated substrings of the shape names to drop
    SHAPES = [s for s in SHAPES if not any(k in s[0] for k in os.environ["SKIP"].split(","))]
if os.environ.get("ONLY"):                                   # comma-separated substrings of the shape names to keep
    SHAPES = [s for s in SHAPES if any(k in s[0] for k in os.environ["ONLY"].split(","))]


def main():
    lib = sda.lib()
    dev = "cuda"
    print(f"{'shape':24s} " + " ".join(f"v{v}: us / TF/s      " for v in VARIANTS))
    for name, M, N, K, conv, act in SHAPES:
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        bias = torch.randn(N, device=dev)
        if conv:
            H, Cin = conv
            a = torch.randn(M // (H * H), H, H, Cin, device=dev).to(BF)
            kw = dict(conv=dict(Hs=H, Ws=H, Cin=Cin, Ho=H, Wo=H))
        else:
            a = torch.randn(M, K, device=dev).to(BF)
            kw = {}
        times = {v: [] for v in VARIANTS}
        for rnd in range(7):
            for v in VARIANTS:
                lib.sdn_debug_set_gemm_variant(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if rnd == 0:
                    ops.gemm(a, w, bias=bias, act=act, **kw)
                e0.record()
                for _ in range(3):
                    ops.gemm(a, w, bias=bias, act=act, **kw)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / 3 * 1e3)
        fl = 2.0 * M * N * K
        cells = []
        for v in VARIANTS:
            t = sorted(times[v])[len(times[v]) // 2]
            cells.append(f"{t:9.1f} {fl / t / 1e6:7.1f}")
        print(f"{name:24s} " + "   ".join(cells))
    lib.sdn_debug_set_gemm_variant(0)


if __name__ == "__main__":
    main()
