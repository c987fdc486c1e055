#!/usr/bin/env python3
"""What would a precise GEMM at ~2 x the 16-bit cost look like?  (first row: groundwork for a later round; CPU, numpy.)

The bf16x3 plan spends three bf16 MFMAs per product: a w ~ a_h w_h + a_l w_h + a_h w_l with a = a_h + a_l, bf16 parts.  Its two
correction products are 2^-8 of the result, so they need the full 8 bits of a bf16 operand.  With an fp16 main term the corrections
are only 2^-11 of the result: 7-8 bits of THEM keep 2^-18.  Scheme emulated here ("h8"):
    a_h = fp16(a), a_l = a - a_h;  S_a[m] = max_k |a_h[m, k]| / 127 (one scale per row);  qa_h = round(a_h / S_a), qa_l = round(a_l / (2^-11 S_a))
    (same for w with one scale per output channel) -- all four int8;  |a_l| <= 2^-11 |a_h| elementwise, so qa_l fits.
    a w  ~  fp16 GEMM (f32 accumulation)  +  2^-11 S_a[m] S_w[n] * int8 GEMM over [qa_l | qa_h] . [qw_h | qw_l]   (i32 accumulation, K' = 2 K)
Both correction terms share ONE scale pair, so they are one int8 GEMM: cost 1 (fp16) + 2 K / 2 (i8 runs at twice the bf16 rate) = 2.0
against 3.0 for bf16x3.  fp8 (e4m3: 4 significant bits) in the same place is emulated too.
Prints the relative error of each scheme against float64 on GEMM shapes of the UNet with activations that have outlier channels."""
import numpy as np


def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def fp16(x):
    return x.astype(np.float16).astype(np.float32)


def e4m3(x):                                   # round to 4 significant bits (the exponent range is not the point here)
    m, e = np.frexp(x)
    return np.ldexp(np.round(m * 16) / 16, e)


def rel(c, ref):
    return float(np.linalg.norm(c - ref) / np.linalg.norm(ref))


def main():
    rng = np.random.default_rng(0)
    print(f"{'shape':28s} {'fp16':>9s} {'bf16':>9s} {'bf16x3':>9s} {'h8 (int8)':>10s} {'h8 (fp8)':>9s} {'h8 main only':>13s}")
    for name, M, N, K in (("conv 320->320 (K=2880)", 512, 320, 2880), ("qkv 320", 512, 960, 320), ("ff1 640", 256, 5120, 640),
                          ("ff2 1280 (K=6400)", 128, 1280, 6400)):
        a = rng.standard_normal((M, K)).astype(np.float32)
        a[:, rng.integers(0, K, K // 64)] *= 12.0                          # outlier channels, as GroupNorm / LayerNorm outputs have
        a *= np.exp(rng.standard_normal((M, 1)) * 0.5).astype(np.float32)   # rows of different magnitude
        w = (rng.standard_normal((N, K)) * K ** -0.5).astype(np.float32)
        ref = a.astype(np.float64) @ w.astype(np.float64).T
        f32mm = lambda x, y: (x.astype(np.float64) @ y.astype(np.float64).T)   # products of the rounded operands, exact sums
        r16 = rel(f32mm(fp16(a), fp16(w)), ref)
        rb = rel(f32mm(bf16(a), bf16(w)), ref)
        ah, wh = bf16(a), bf16(w)
        al, wl = bf16(a - ah), bf16(w - wh)
        rx3 = rel(f32mm(ah, wh) + f32mm(al, wh) + f32mm(ah, wl), ref)
        ah, wh = fp16(a), fp16(w)
        al, wl = a - ah, w - wh
        sa = np.abs(ah).max(1, keepdims=True) / 127.0
        sw = np.abs(wh).max(1, keepdims=True) / 127.0
        qah, qwh = np.round(ah / sa), np.round(wh / sw)
        qal, qwl = np.round(al / (2.0 ** -11 * sa)), np.round(wl / (2.0 ** -11 * sw))
        assert np.abs(qal).max() <= 127 and np.abs(qwl).max() <= 127
        corr = 2.0 ** -11 * sa.astype(np.float64) * sw.astype(np.float64).T * (f32mm(qal, qwh) + f32mm(qah, qwl))
        rh8 = rel(f32mm(ah, wh) + corr, ref)
        corr8 = f32mm(e4m3(al), e4m3(wh)) + f32mm(e4m3(ah), e4m3(wl))
        rf8 = rel(f32mm(ah, wh) + corr8, ref)
        print(f"{name:28s} {r16:9.2e} {rb:9.2e} {rx3:9.2e} {rh8:10.2e} {rf8:9.2e} {rel(f32mm(ah, wh), ref):13.2e}")


if __name__ == "__main__":
    main()
