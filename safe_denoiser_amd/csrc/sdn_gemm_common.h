// Shared by the GEMM-family translation units (sdn_gemm.hip, sdn_ffn.hip): launch arguments, epilogue activations, the
// LDS tile addressing and the buffer-resource helpers.  See sdn_gemm.hip for the design notes.
#pragma once
#include "sdn_common.h"
#include "sdn_ops.h"

namespace sdn_gemm_detail {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int BK = 64;

struct GemmArgs {
  const __bf16* a;  const __bf16* a2;  const __bf16* w;
  const float* bias;  const float* rowbias;  const float* rowgate;  const __bf16* residual;  void* out;
  int M, N, K, K1;
  int a_mode, Hs, Ws, Cin, Ho, Wo, stride, upsample, conv_off;
  int act, out_kind, rows_per_batch, ld_rowbias, ld_rowgate, residual_bcast, n_valid, ldc;
  int tiles_m, tiles_n;
  int panel;                 // tile order inside an XCD's share: 0 = row-major, else n-tiles per panel (see k_gemm_dma)
  const float* ln_c; const float* ln_d; float ln_eps; const float* ln_stats;
  float* col_stats;          // per 128-row block and output column: (sum, sum of squares) of the stored 16-bit values, for the
                             // GroupNorm that consumes this tensor (sdn_gemm_stats_* / sdn_groupnorm_cols_*); nullptr = off   // LNF kernels: LayerNorm folded into this GEMM (see k_gemm_dma)
  int kt_per_split;          // split-K: k-tiles per blockIdx.y slice (0 = no split); each slice writes its own fp32 partial
  long split_stride;         // bytes between the partial outputs of consecutive slices
  int res_lds;               // residual goes through the LDS staging slab (16-bit staged output, offsets fit 31 bits)
  unsigned res_bytes;        // buffer size of the residual for the DMA's bounds check
  const __bf16* res_pre;     // 16-bit residual [M, ldc] added into the ACCUMULATORS before the k loop (sdn_gemm_desc.res_pre): its loads
                             // ride in the shadow of the first k-tile's DMA and the epilogue is the lean no-residual one
  int x3_out;                // bf16x3 plan (operands = bf16 hi|lo|hi triples, K = 3 x the logical K; sdn_gemm_x3t): the residual is F32
                             // [M, ldc], added into the accumulators before the k loop, and the output goes straight from the
                             // registers to global memory: 1 = f32 [M, ldc]; 2 = GEGLU, triple [M, 3 ldc]; 3 = triple [M, 3 ldc]
  int h8_t16;                // experimental "h8" operand form (sdn_gemm_desc.x3_out = 5, fp16 instance): the first h8_t16 k-tiles are fp16,
                             // the rest are e4m3 bytes in two equal segments (corrections); 0 = off
  unsigned long long* stamps; // diagnostics: 8 s_memtime stamp slots per workgroup (tools/gemm_stamps.py); nullptr in production
  int dbg;                   // timing-only ablations (tools/bench_gemm.py): 1 = no global stores, 2 = DMA only for k-tile 0
};

__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 p = {(__bf16)lo, (__bf16)hi};                 // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
  return *reinterpret_cast<unsigned*>(&p);
}
__device__ __forceinline__ float silu_f(float v) { return v / (1.f + __expf(-v)); }
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below one 16-bit output rounding): one v_exp, one v_rcp and
// five FMAs instead of libm's branchy erff -- the GEGLU epilogue evaluates it 1280..5120 times per output row.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}
// x = hi + lo to 2^-17 relative (16 mantissa bits): hi = bf16(x), lo = bf16(x - hi); packed pairs of four values
__device__ __forceinline__ void split_hi_lo4(const f32x4 v, uint2& hi, uint2& lo) {
  hi.x = pack_bf16(v[0], v[1]); hi.y = pack_bf16(v[2], v[3]);
  lo.x = pack_bf16(v[0] - __uint_as_float(hi.x << 16), v[1] - __uint_as_float(hi.x & 0xffff0000u));
  lo.y = pack_bf16(v[2] - __uint_as_float(hi.y << 16), v[3] - __uint_as_float(hi.y & 0xffff0000u));
}
__device__ __forceinline__ float gelu_erf_ref(float v) { return 0.5f * v * (1.f + erf_as(v * 0.70710678118654752f)); }
// GELU (erf form) for a 16-BIT output: x * Phi(x) with Phi(x) = 1 / (1 + 2^(x (a + b x^2 + c x^4))), a quintic-argument logistic
// fitted to the exact function (minimax over |x| <= 12): |error| <= 2.6e-5 absolute, <= 7.7e-5 relative for x > 0.02 -- 1/25 of
// the bf16 and 1/3 of the fp16 half-ulp of the stored result -- in 9 VALU instructions (2 transcendental) instead of 17.
// In-kernel timing made the point: at K = 320 the GEGLU projection's epilogue (32 GELUs per lane) costs as much as its whole
// k loop (the per-output cost equals ~310 columns of K), and that epilogue is mostly this function.  x^2 is clamped at 100:
// beyond |x| = 11 the quintic would change sign; the clamped argument keeps growing linearly, so the tails are exact 0 / x.
__device__ __forceinline__ float gelu_erf(float v) {
  const float x2 = fminf(v * v, 100.0f);
  const float u = v * fmaf(x2, fmaf(x2, 0.0010142630198970437f, -0.10677572339773178f), -2.301121234893799f);   // -log2(e) folded in
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u));
}
__device__ __forceinline__ float gelu_tanh(float v) {                       // GELU(approximate="tanh"), MMDiT feed-forward
  const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
  // 0.5 (1 + tanh u) = 1 / (1 + exp(-2u))   (one v_exp + one v_rcp; saturates cleanly for |u| large)
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * u));
}

__device__ __forceinline__ float quick_gelu(float v) {                      // x * sigmoid(1.702 x), CLIP text encoder
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * v));
}

// byte offset of 16-B chunk c of row r inside a [rows][8 chunks] tile
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

// ================================================================================================
// Both operands are staged by LDS-DMA
// (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write pass, and the conv halo / M tail is zero-filled
// by the buffer descriptor's range check (out-of-range voffset -> the DMA writes zeros).
// One wave-instruction writes 1 KiB = 8 rows x 128 B of the tile linearly, so the chunk ^ (row & 7) swizzle
// is applied to the per-lane SOURCE address (lane (row = l>>3, slot = l&7) fetches logical chunk slot ^ row&7)
// and to the fragment reads -- never to the LDS destination (guide, rule 21).
// ================================================================================================
#if defined(__HIP_DEVICE_COMPILE__)     // buffer-resource builtins exist only in the device pass
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#endif

// sdn_conv.hip: 3x3 stride-1 convolution with the A operand served from an LDS slab ring; returns SDN_GEMM_NOT_SLAB when the
// problem does not suit it (the caller then launches the implicit-GEMM kernel).  Same bits either way.
constexpr int SDN_GEMM_NOT_SLAB = -1001;
int dispatch_conv_slab(int dtype, const GemmArgs& g, hipStream_t st);

}  // namespace sdn_gemm_detail
