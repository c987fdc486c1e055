#!/usr/bin/env python3
"""Kernel-level price and accuracy of the experimental "h8" precise GEMM: fp16 main term + two e4m3 correction
products by operand expansion on the 256-row LDS-DMA tile (`sdn_gemm_f16` with `x3_out = 5`, kernel `k_gemm_h8`) against
  * the fp16 GEMM (`sdn_gemm_f16`, f32 output: the same epilogue),
  * the bf16x3 GEMM by operand expansion (`sdn_gemm_bf16`, `x3_out = 1`, K' = 3 K) -- what the tolerance-meeting plan runs today,
on GEMM shapes of the UNet at the benchmark batch; in-process, interleaved rounds, median.  Accuracy: rel L2 against a float64
product on a row sample.  The operand rows are built with torch here (a producer kernel would write them):
    A' = [fp16(a) | e4m3(2^11 (a - fp16(a))) | e4m3(a)]      W' = [fp16(w) | e4m3(w) | e4m3(2^11 (w - fp16(w)))]     (4 bytes per element)
    python tools/bench_h8_gemm.py          (B = 192 by default: env B)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_denoiser_amd as sda  # noqa: E402
from safe_denoiser_amd import _lib  # noqa: E402

B = int(os.environ.get("B", "192"))
SHAPES = [("ff2 1280 @16", B * 256, 1280, 5120), ("qkv 1280 @16", B * 256, 3840, 1280), ("proj 1280 @16", B * 256, 1280, 1280),
          ("ff1 640 @32 (no GEGLU)", B * 1024, 5120, 640), ("K=3072 N=1280 (conv-sized)", B * 256, 1280, 3072),
          ("ff2 640 @32 (N % 256 != 0: 320-wide tile, spills)", B * 1024, 640, 2560)]
if os.environ.get("ONLY"):
    SHAPES = [s for s in SHAPES if any(k in s[0] for k in os.environ["ONLY"].split(","))]
dev = "cuda"
lib = sda.lib()


def desc(M, N, K, x3=0, out_kind=1):
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.out_kind, d.x3_out, d.ldc = M, N, K, out_kind, x3, N
    return d


def run(fn, d, a, w, out):
    _lib.check(fn(C.byref(d), a.data_ptr(), None, w.data_ptr(), None, None, None, None, out.data_ptr(), _lib.stream_ptr()), "gemm")


def e4m3(x):
    return x.to(torch.float8_e4m3fn).view(torch.uint8)


def timeit(f, rounds, key, acc):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        f()
    e1.record()
    torch.cuda.synchronize()
    acc.setdefault(key, []).append(e0.elapsed_time(e1) / 3 * 1e3)


def main():
    print(f"B = {B}; times in us (median of 7 interleaved rounds of 3), TFLOP/s algorithmic; rel L2 vs float64 on 2048 rows")
    print(f"{'shape':32s} {'fp16':>16s} {'bf16x3 (x3t)':>22s} {'h8':>22s}   accuracy fp16 / bf16x3 / h8")
    for name, M, N, K in SHAPES:
        M = min(M, ((2 ** 31 - 1) // (6 * K)) // 256 * 256)            # one launch addresses an operand with 31-bit offsets (the plan chunks rows)
        g = torch.Generator(device=dev).manual_seed(1)
        a = torch.randn(M, K, device=dev, generator=g)
        a[:, ::64] *= 12.0                                              # outlier channels
        w = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
        ah, wh = a.half(), w.half()
        al, wl = a - ah.float(), w - wh.float()
        a8 = torch.cat([ah.view(torch.uint8), e4m3(al * 2048.0), e4m3(ah.float())], dim=1).contiguous()       # [M, 4K] bytes
        w8 = torch.cat([wh.view(torch.uint8), e4m3(wh.float()), e4m3(wl * 2048.0)], dim=1).contiguous()
        ab, wb = a.bfloat16(), w.bfloat16()
        abl, wbl = (a - ab.float()).bfloat16(), (w - wb.float()).bfloat16()
        a3 = torch.cat([ab, abl, ab], dim=1).contiguous()               # [hi | lo | hi]
        w3 = torch.cat([wb, wb, wbl], dim=1).contiguous()               # [hi | hi | lo]
        del al, wl, abl, wbl
        o16, o3, o8 = (torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(3))
        d16, d3, d8 = desc(M, N, K), desc(M, N, 3 * K, x3=1), desc(M, N, 2 * K, x3=5)
        f16 = lambda: run(lib.sdn_gemm_f16, d16, ah, wh, o16)
        f3 = lambda: run(lib.sdn_gemm_bf16, d3, a3, w3, o3)
        f8 = lambda: run(lib.sdn_gemm_f16, d8, a8, w8, o8)
        for f in (f16, f3, f8):
            f()
        torch.cuda.synchronize()
        t = {}
        for _ in range(7):
            timeit(f16, 1, "16", t); timeit(f3, 1, "x3", t); timeit(f8, 1, "h8", t)
        med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
        rows = torch.arange(0, M, max(1, M // 2048), device=dev)[:2048]
        ref = a[rows].double() @ w.double().T
        rel = lambda o: float((o[rows].double() - ref).norm() / ref.norm())
        fl = 2.0 * M * N * K
        cell = lambda k: f"{med[k]:8.1f} {fl / med[k] / 1e6:6.0f}"
        print(f"{name + f' M={M}':44s} {cell('16')}   {cell('x3')} x{med['x3'] / med['16']:.2f}   {cell('h8')} x{med['h8'] / med['16']:.2f}   "
              f"{rel(o16):.2e} / {rel(o3):.2e} / {rel(o8):.2e}", flush=True)
        del a, w, ah, wh, a8, w8, a3, w3, ab, wb, o16, o3, o8
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
