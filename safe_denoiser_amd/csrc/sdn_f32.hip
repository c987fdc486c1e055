// fp32 storage mode of the denoiser-network operators (dtype 2 of sdn_unet_config): every activation and every weight is
// IEEE fp32 and every contraction runs on the f32-input matrix cores (v_mfma_f32_16x16x4_f32: exact f32 products, f32
// accumulation, bit-for-bit a k-ordered fmaf chain -- guide section 3 "FP32-input MFMA").  1/16 of the bf16 MFMA rate, so
// this is the plan's PRECISION mode, not its throughput mode: it exists so that the same launch plan (same wiring, same
// fusions, same weight manifest) can be checked against the pure-fp32 reference arithmetic at full size to ~1e-6 per
// forward, where 16-bit storage cannot do better than 1e-3 (fp16) / 1e-2 (bf16) whatever the implementation
// (north star: latents <= 1e-3 rel).  Same operators and argument meaning as the 16-bit entry points of include/sdn.h:
//   sdn_gemm_f32          <-> sdn_gemm_bf16        (plain / two-source / implicit 3x3 conv A operand, every epilogue)
//   sdn_groupnorm_f32     <-> sdn_groupnorm_bf16   (optional channel concat of two maps, optional SiLU)
//   sdn_layernorm_f32     <-> sdn_layernorm_bf16
//   sdn_attention_f32     <-> sdn_attention_bf16   (flash form: scores never leave the registers)
//   sdn_conv_in_f32, sdn_timestep_embed_f32
// Kernels here favour plainness over the last 2x: tiles are small, staging is register -> LDS with one prefetched k-tile.
#include <math.h>
#include <stdlib.h>
#include <type_traits>

#include "sdn_common.h"
#include "sdn_ops.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float silu_f(float v) { return v / (1.f + expf(-v)); }
__device__ __forceinline__ float gelu_erf_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_tanh_f(float v) {
  return 0.5f * v * (1.f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
}
__device__ __forceinline__ float quick_gelu_f(float v) { return v / (1.f + expf(-1.702f * v)); }

// bf16 hi | lo | hi TRIPLE form of an f32 tensor (include/sdn.h, "bf16x3 by operand expansion"): element (row, c) of a
// [rows, C] tensor lives at out16[row * 3C + c] (hi), [.. + C] (lo), [.. + 2C] (hi again)
__device__ __forceinline__ unsigned short bf16_rne(float v) {
  __bf16 h = (__bf16)v;
  return *reinterpret_cast<unsigned short*>(&h);
}
__device__ __forceinline__ void store_triple1(unsigned short* row3, int C, int c, float v) {
  const unsigned short hi = bf16_rne(v);
  const unsigned short lo = bf16_rne(v - __uint_as_float((unsigned)hi << 16));
  row3[c] = hi; row3[C + c] = lo; row3[2 * C + c] = hi;
}
__device__ __forceinline__ void store_triple2(unsigned short* row3, int C, int c, float v0, float v1) {     // c even
  const unsigned h0 = bf16_rne(v0), h1 = bf16_rne(v1);
  const unsigned l0 = bf16_rne(v0 - __uint_as_float(h0 << 16)), l1 = bf16_rne(v1 - __uint_as_float(h1 << 16));
  const unsigned hi = h0 | (h1 << 16), lo = l0 | (l1 << 16);
  *reinterpret_cast<unsigned*>(row3 + c) = hi;
  *reinterpret_cast<unsigned*>(row3 + C + c) = lo;
  *reinterpret_cast<unsigned*>(row3 + 2 * C + c) = hi;
}

// ================================================================================================
// GEMM / implicit conv:  C[M,N] = A[M,K] . W[N,K]^T, 64 x 64 tile, BK = 16, 4 waves (one per SIMD); wave w owns rows
// 16w..16w+15 of the tile and all 64 columns = 4 accumulators of v_mfma_f32_16x16x4_f32 in the SWAPPED orientation
// (weights = A operand, activations = B operand): a lane ends with 4 consecutive output columns of one row.
// LDS images are k-major ([16][64 + 16]): the fragment reads of a wave (16 consecutive rows x 4 k) are conflict-free.
// ================================================================================================
constexpr int BM = 64, BN = 64, BK = 16, LDT = 80;

struct GemmArgsF {
  const float* a; const float* a2; const float* w;
  const float* bias; const float* rowbias; const float* rowgate; const float* residual; float* out;
  int M, N, K, K1;
  int a_mode, Hs, Ws, Cin, Ho, Wo, stride, upsample, conv_off;
  int act, out_kind, rows_per_batch, ld_rowbias, ld_rowgate, residual_bcast, n_valid, ldc;
  int tiles_n;
  int dbg;                    // timing-only ablations of k_gemm_x3 (SDN_X3_DBG; tools/bench_x3_gemm.py): 1 no split arithmetic, 2 no global loads after the prologue, 4 no LDS writes, 8 one MFMA per product
};

// One output row's share of a wave's 16 x 64 accumulator block: acc[j][e] <-> row m, column n0 + 16 j + 4 fq + e.  Bias,
// per-sample row bias / gate, residual, activation (incl. the interleaved GEGLU pairs) and the three output layouts.
template <int NJ = 4>
__device__ __forceinline__ void gemm_row_epilogue(const GemmArgsF& g, const f32x4 (&acc)[NJ], const int m, const int n0, const int fq) {
  if (m >= g.M) return;
  const int b = g.rows_per_batch > 0 ? m / g.rows_per_batch : 0;
  if (g.act == SDN_ACT_GEGLU) {                                              // blocks of 16 value columns, then their 16 gates
#pragma unroll
    for (int j = 0; j + 1 < NJ; j += 2) {
      const int n = n0 + j * 16 + fq * 4;
      if (n >= g.N) continue;
      f32x4 hv = acc[j], gv = acc[j + 1];
      if (g.bias) { hv += *reinterpret_cast<const f32x4*>(g.bias + n); gv += *reinterpret_cast<const f32x4*>(g.bias + n + 16); }
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = hv[e] * gelu_erf_f(gv[e]);
      const int nc = ((n0 + j * 16) >> 1) + fq * 4;
      *reinterpret_cast<f32x4*>(g.out + (long)m * g.ldc + nc) = o;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = n0 + j * 16 + fq * 4;
    if (n >= g.N) continue;
    f32x4 v = acc[j];
    if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + n);
    if (g.rowbias) v += *reinterpret_cast<const f32x4*>(g.rowbias + (long)b * g.ld_rowbias + n);
    if (g.rowgate) v *= *reinterpret_cast<const f32x4*>(g.rowgate + (long)b * g.ld_rowgate + n);
    if (g.residual) {
      const long rrow = g.residual_bcast ? (long)(m - b * g.rows_per_batch) : (long)m;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < g.n_valid) v[e] += g.residual[rrow * g.ldc + n + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = v[e];
      if (g.act == SDN_ACT_SILU) t = silu_f(t);
      else if (g.act == SDN_ACT_GELU_TANH) t = gelu_tanh_f(t);
      else if (g.act == SDN_ACT_QUICK_GELU) t = quick_gelu_f(t);
      if (n + e >= g.n_valid) continue;
      if (g.out_kind == SDN_OUT_F32_NCHW) {
        const int p = m - b * g.rows_per_batch;
        g.out[((long)b * g.n_valid + n + e) * g.rows_per_batch + p] = t;
      } else {
        g.out[(long)m * g.ldc + n + e] = t;
      }
    }
  }
}

__global__ void __launch_bounds__(256)
k_gemm_f32(const GemmArgsF g) {
  __shared__ float sa[BK * LDT], sw[BK * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int tm = blockIdx.x / g.tiles_n, tn = blockIdx.x - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  // staging role: thread -> (row = tid / 4, 4 consecutive k = (tid % 4) * 4) of the A tile and of the W tile
  const int srow = tid >> 2, sk = (tid & 3) * 4;
  const int am = m0 + srow;
  const bool a_ok = am < g.M;
  const int wn = n0 + srow;
  const bool w_ok = wn < g.N;
  // im2col: this thread's output pixel
  int pb = 0, cy = 0, cx = 0;
  if (g.a_mode == 1 && a_ok) {
    const int hw = g.Ho * g.Wo;
    pb = am / hw;
    const int p = am - pb * hw;
    const int oy = p / g.Wo, ox = p - oy * g.Wo;
    cy = oy * g.stride + g.conv_off; cx = ox * g.stride + g.conv_off;      // tap centre in the (virtual, upsampled) input
  }
  const int Hi = g.upsample ? 2 * g.Hs : g.Hs, Wi = g.upsample ? 2 * g.Ws : g.Ws;
  const int ld1 = g.K1, ld2 = g.K - g.K1;

  auto load_a = [&](int k0) -> f32x4 {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!a_ok) return v;
    if (g.a_mode == 1) {                                                     // weights are [N][ky][kx][Cin]: k -> (tap, channel)
      const int tap = k0 / g.Cin, c0 = k0 - tap * g.Cin;
      const int y = cy + tap / 3 - 1, x = cx + tap % 3 - 1;
      if (y < 0 || y >= Hi || x < 0 || x >= Wi) return v;
      const int sy = g.upsample ? y >> 1 : y, sx = g.upsample ? x >> 1 : x;
      return *reinterpret_cast<const f32x4*>(g.a + (((long)pb * g.Hs + sy) * g.Ws + sx) * g.Cin + c0 + sk);
    }
    if (k0 < g.K1) return *reinterpret_cast<const f32x4*>(g.a + (long)am * ld1 + k0 + sk);
    return *reinterpret_cast<const f32x4*>(g.a2 + (long)am * ld2 + (k0 - g.K1) + sk);
  };
  auto load_w = [&](int k0) -> f32x4 {
    if (!w_ok) return (f32x4){0.f, 0.f, 0.f, 0.f};
    return *reinterpret_cast<const f32x4*>(g.w + (long)wn * g.K + k0 + sk);
  };

  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 ra = load_a(0), rw = load_w(0);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { sa[(sk + e) * LDT + srow] = ra[e]; sw[(sk + e) * LDT + srow] = rw[e]; }
    __syncthreads();
    if (k0 + BK < g.K) { ra = load_a(k0 + BK); rw = load_w(k0 + BK); }       // next k-tile in flight under the MFMAs
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      const float bv = sa[(ks * 4 + fq) * LDT + wid * 16 + fr];              // B[k = fq][j = fr] = A_tile[row 16w + fr][k]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float av = sw[(ks * 4 + fq) * LDT + j * 16 + fr];              // A[i = fr][k = fq] = W_tile[col 16j + fr][k]
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: acc[j][e] <-> row m0 + 16w + fr, column n0 + 16j + 4fq + e ----
  gemm_row_epilogue(g, acc, m0 + wid * 16 + fr, n0, fq);
}

// ================================================================================================
// GroupNorm (+ SiLU, + channel concat of two maps): one workgroup per (sample, group); statistics are accumulated in
// double (two maps of 64 x 64 x 10..80 values per group: a plain f32 sum / sum of squares would cancel), then the apply pass.
// ================================================================================================
__global__ void __launch_bounds__(256)
k_groupnorm_f32(const float* __restrict__ x, const float* __restrict__ x2, int hw, int c1, int c2, int groups, float eps,
                int silu, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ out, int triple) {
  __shared__ double red[2][4];
  __shared__ float stat[2];
  const int C = c1 + c2, cpg = C / groups;
  const int b = blockIdx.x / groups, gi = blockIdx.x - b * groups;
  const int ch0 = gi * cpg;
  // thread -> (channel pair tx of the group, pixel ty + k * ppp): float2 accesses, no per-element index arithmetic
  // (cpg and c1 are even for every layer of the plan: 10 ... 80 channels per group)
  const int hp = cpg >> 1, ppp = 256 / hp;
  const int tx = threadIdx.x % hp, ty = threadIdx.x / hp;
  const bool live = ty < ppp;
  const int c = ch0 + 2 * tx;
  const float* src = c < c1 ? x + (long)b * hw * c1 + c : x2 + (long)b * hw * c2 + (c - c1);
  const int ld = c < c1 ? c1 : c2;
  double s = 0.0, q = 0.0;
  if (live)
    for (int p = ty; p < hw; p += ppp) {
      const float2 v = *reinterpret_cast<const float2*>(src + (long)p * ld);
      const double a0 = v.x, a1 = v.y;
      s += a0 + a1; q += a0 * a0 + a1 * a1;
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double n = (double)hw * cpg;
    const double ts = red[0][0] + red[0][1] + red[0][2] + red[0][3], tq = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double mean = ts / n;
    double var = tq / n - mean * mean;                                       // biased variance, as torch's group_norm
    if (var < 0.0) var = 0.0;
    stat[0] = (float)mean; stat[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  if (!live) return;
  const float mean = stat[0], rstd = stat[1];
  const float ga0 = gamma[c], ga1 = gamma[c + 1], be0 = beta[c], be1 = beta[c + 1];
  float* dst = out + (long)b * hw * C + c;
  for (int p = ty; p < hw; p += ppp) {
    const float2 v = *reinterpret_cast<const float2*>(src + (long)p * ld);
    float o0 = (v.x - mean) * rstd * ga0 + be0, o1 = (v.y - mean) * rstd * ga1 + be1;      // centred first: |mean| >> std must not cancel
    if (silu) { o0 = silu_f(o0); o1 = silu_f(o1); }
    if (triple) store_triple2(reinterpret_cast<unsigned short*>(out) + ((long)b * hw + p) * 3 * C, C, c, o0, o1);
    else *reinterpret_cast<float2*>(dst + (long)p * C) = make_float2(o0, o1);
  }
}

// general form (odd channels per group, or an odd split of a concatenated input): one element per access
__global__ void __launch_bounds__(256)
k_groupnorm_f32_any(const float* __restrict__ x, const float* __restrict__ x2, int hw, int c1, int c2, int groups, float eps,
                    int silu, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ out, int triple) {
  __shared__ double red[2][4];
  __shared__ float stat[2];
  const int C = c1 + c2, cpg = C / groups;
  const int b = blockIdx.x / groups, gi = blockIdx.x - b * groups;
  const int ch0 = gi * cpg;
  const long n = (long)hw * cpg;
  auto at = [&](long e) -> float {
    const int p = (int)(e / cpg), c = ch0 + (int)(e - (long)p * cpg);
    return c < c1 ? x[((long)b * hw + p) * c1 + c] : x2[((long)b * hw + p) * c2 + (c - c1)];
  };
  double s = 0.0, q = 0.0;
  for (long e = threadIdx.x; e < n; e += 256) { const double v = at(e); s += v; q += v * v; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ts = red[0][0] + red[0][1] + red[0][2] + red[0][3], tq = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double mean = ts / (double)n;
    double var = tq / (double)n - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[0] = (float)mean; stat[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
  for (long e = threadIdx.x; e < n; e += 256) {
    const int p = (int)(e / cpg), c = ch0 + (int)(e - (long)p * cpg);
    float v = (at(e) - mean) * rstd * gamma[c] + beta[c];
    if (silu) v = silu_f(v);
    if (triple) store_triple1(reinterpret_cast<unsigned short*>(out) + ((long)b * hw + p) * 3 * C, C, c, v);
    else out[((long)b * hw + p) * C + c] = v;
  }
}

// ---- GroupNorm, row-major form (round 4): whole 4C-byte rows are read (float4 per lane, coalesced) instead of one group's
// 40 ... 320-byte slice per pixel.  A sample's [hw, C] matrix is cut into `nchunk` row chunks; pass 1 leaves one (sum, sum of
// squares) pair per (sample, chunk, group) in DOUBLE (|mean| >> std must not cancel), reduced in a fixed order; pass 2 sums the
// chunk partials (fixed order), normalises its chunk and writes f32 rows or the bf16 hi|lo|hi triple.  8 + 4 (or 6) bytes per
// element instead of three strided passes.
constexpr int GN_MAXC = 2560;
__global__ void __launch_bounds__(256)
k_gn_rows_stats(const float* __restrict__ x, const float* __restrict__ x2, int hw, int c1, int c2, int groups, int rows_per_chunk,
                double* __restrict__ part) {
  __shared__ double ls[GN_MAXC], lq[GN_MAXC];
  const int C = c1 + c2, cq = C / 4, cpg = C / groups;
  const int nchunk = (hw + rows_per_chunk - 1) / rows_per_chunk;
  const int b = blockIdx.x / nchunk, ch = blockIdx.x - b * nchunk;
  const int r0 = ch * rows_per_chunk, r1 = min(hw, r0 + rows_per_chunk);
  // thread -> column quads qd, qd + 256, ... (all rows of the chunk): per-column sums stay in one thread, no cross-thread order
  for (int qd = threadIdx.x; qd < cq; qd += 256) {
    const int c = qd * 4;
    const float* src = c < c1 ? x + ((long)b * hw) * c1 + c : x2 + ((long)b * hw) * c2 + (c - c1);
    const int ld = c < c1 ? c1 : c2;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)r * ld);
      const double a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
      q0 = fma(a0, a0, q0); q1 = fma(a1, a1, q1); q2 = fma(a2, a2, q2); q3 = fma(a3, a3, q3);
    }
    ls[c] = s0; ls[c + 1] = s1; ls[c + 2] = s2; ls[c + 3] = s3;
    lq[c] = q0; lq[c + 1] = q1; lq[c + 2] = q2; lq[c + 3] = q3;
  }
  __syncthreads();
  if (threadIdx.x < groups) {
    double s = 0, q = 0;
    for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) { s += ls[c]; q += lq[c]; }
    double* o = part + (((long)b * nchunk + ch) * groups + threadIdx.x) * 2;
    o[0] = s; o[1] = q;
  }
}

__global__ void __launch_bounds__(256)
k_gn_rows_apply(const float* __restrict__ x, const float* __restrict__ x2, int hw, int c1, int c2, int groups, float eps, int silu,
                const float* __restrict__ gamma, const float* __restrict__ beta, int rows_per_chunk, const double* __restrict__ part,
                float* __restrict__ out, int triple) {
  __shared__ float smean[64], srstd[64];
  const int C = c1 + c2, cq = C / 4, cpg = C / groups;
  const int nchunk = (hw + rows_per_chunk - 1) / rows_per_chunk;
  const int b = blockIdx.x / nchunk, ch = blockIdx.x - b * nchunk;
  if (threadIdx.x < groups) {
    double s = 0, q = 0;
    for (int k = 0; k < nchunk; ++k) {
      const double* pp = part + (((long)b * nchunk + k) * groups + threadIdx.x) * 2;
      s += pp[0]; q += pp[1];
    }
    const double n = (double)hw * cpg, mean = s / n;
    double var = q / n - mean * mean;                                        // biased variance, as torch's group_norm
    if (var < 0.0) var = 0.0;
    smean[threadIdx.x] = (float)mean; srstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int r0 = ch * rows_per_chunk, r1 = min(hw, r0 + rows_per_chunk);
  for (int qd = threadIdx.x; qd < cq; qd += 256) {
    const int c = qd * 4;
    const float* src = c < c1 ? x + ((long)b * hw) * c1 + c : x2 + ((long)b * hw) * c2 + (c - c1);
    const int ld = c < c1 ? c1 : c2;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    float mu[4], rs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int g = (c + e) / cpg; mu[e] = smean[g]; rs[e] = srstd[g]; }
#pragma unroll 4
    for (int r = r0; r < r1; ++r) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)r * ld);
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (v[e] - mu[e]) * rs[e] * ga[e] + be[e];                       // centred first: |mean| >> std must not cancel
        if (silu) o[e] = silu_f(o[e]);
      }
      const long row = (long)b * hw + r;
      if (triple) {
        unsigned short* row3 = reinterpret_cast<unsigned short*>(out) + row * 3 * C;
        const unsigned h0 = bf16_rne(o[0]), h1 = bf16_rne(o[1]), h2 = bf16_rne(o[2]), h3 = bf16_rne(o[3]);
        const unsigned l0 = bf16_rne(o[0] - __uint_as_float(h0 << 16)), l1 = bf16_rne(o[1] - __uint_as_float(h1 << 16));
        const unsigned l2 = bf16_rne(o[2] - __uint_as_float(h2 << 16)), l3 = bf16_rne(o[3] - __uint_as_float(h3 << 16));
        const uint2 hi = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16)), lo = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
        *reinterpret_cast<uint2*>(row3 + c) = hi;
        *reinterpret_cast<uint2*>(row3 + C + c) = lo;
        *reinterpret_cast<uint2*>(row3 + 2 * C + c) = hi;
      } else {
        *reinterpret_cast<f32x4*>(out + row * C + c) = (f32x4){o[0], o[1], o[2], o[3]};
      }
    }
  }
}

// ================================================================================================
// LayerNorm over the last axis: one wave per row, two passes over registers / L1 (mean, then centred sum of squares).
__global__ void __launch_bounds__(256)
k_layernorm_f32(const float* __restrict__ x, long rows, int c, float eps, const float* __restrict__ gamma,
                const float* __restrict__ beta, float* __restrict__ out, int triple) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + row * c;
  float s = 0.f;
  for (int i = lane; i < c; i += 64) s += xr[i];
  const float mean = wave_sum(s) / (float)c;
  float q = 0.f;
  for (int i = lane; i < c; i += 64) { const float d = xr[i] - mean; q = fmaf(d, d, q); }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)c + eps);
  if (triple) {
    unsigned short* row3 = reinterpret_cast<unsigned short*>(out) + row * 3 * c;
    if ((c & 1) == 0) {
      for (int i = 2 * lane; i < c; i += 128)
        store_triple2(row3, c, i, (xr[i] - mean) * rstd * gamma[i] + beta[i], (xr[i + 1] - mean) * rstd * gamma[i + 1] + beta[i + 1]);
    } else {
      for (int i = lane; i < c; i += 64) store_triple1(row3, c, i, (xr[i] - mean) * rstd * gamma[i] + beta[i]);
    }
    return;
  }
  for (int i = lane; i < c; i += 64) out[row * c + i] = (xr[i] - mean) * rstd * gamma[i] + beta[i];
}

// ================================================================================================
// Attention, flash form on the f32 matrix cores.  Workgroup = 4 waves = 64 queries of one (batch, head); keys / values
// arrive in tiles of 16 through LDS.  Per wave (16 queries):  S^T[key][query] = K . Q^T  (A = K tile, B = Q^T), so a
// lane holds 4 keys' scores of ONE query; the running max / sum of a query live in the 4 lanes {q, q+16, q+32, q+48};
// O^T[dim][query] += V^T . P^T takes register s of the score accumulator directly as the B operand of its k-step s
// (k-step s, lane group g  <->  key 4g + s on both operands).  Exponentials are expf(x - max): the reference's softmax.
// ================================================================================================
// MASK (CLIP text encoder, sdn_masked_attention_f32): causal key <= query and an optional per-sample key-padding mask [B, nk].
template <int HD, bool MASK = false>
__global__ void __launch_bounds__(256)
k_attention_f32(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ out,
                int heads, int nq, int nk, int ldq, int ldk, int ldv, int ldo, float scale, int causal = 0,
                const int* __restrict__ kmask = nullptr) {
  constexpr int DB = (HD + 15) / 16, DP = DB * 16;                          // head dim padded to whole 16-column blocks
  constexpr int KS = HD / 4;                                               // k-steps of the score product (HD % 4 == 0)
  constexpr int LDK = ((HD + 29) / 32) * 32 + 2;                            // rows 2 banks apart mod 32: conflict-free reads
  constexpr int LDV = DP + 4;                                               // rows 4 apart -> 16 banks apart per lane group
  __shared__ float sk[16 * LDK], sv[16 * LDV];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int qblocks = (nq + 63) / 64;
  const int bh = blockIdx.x / qblocks, qb = blockIdx.x - bh * qblocks;
  const int b = bh / heads, h = bh - b * heads;
  const int qi = qb * 64 + wid * 16 + fr;                                  // this lane's query
  const bool q_ok = qi < nq;
  const float* qp = q + ((long)b * nq + (q_ok ? qi : 0)) * ldq + h * HD;
  float qf[KS];                                                            // B operand of k-step s: Q[query fr][dim 4s + fq]
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = q_ok ? qp[4 * s + fq] * scale : 0.f;
  f32x4 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun = -INFINITY, lrun = 0.f;
  for (int i = tid; i < 16 * LDV; i += 256) sv[i] = 0.f;                    // the padding columns HD..DP stay zero
  const float* kb = k + (long)b * nk * ldk + h * HD;
  const float* vb = v + (long)b * nk * ldv + h * HD;
  for (int k0 = 0; k0 < nk; k0 += 16) {
    __syncthreads();                                                       // previous tile fully consumed
    for (int i = tid; i < 16 * (HD / 4); i += 256) {
      const int r = i / (HD / 4), c4 = (i - r * (HD / 4)) * 4;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (k0 + r < nk) {
        kv = *reinterpret_cast<const f32x4*>(kb + (long)(k0 + r) * ldk + c4);
        vv = *reinterpret_cast<const f32x4*>(vb + (long)(k0 + r) * ldv + c4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { sk[r * LDK + c4 + e] = kv[e]; sv[r * LDV + c4 + e] = vv[e]; }
    }
    __syncthreads();
    f32x4 st = {0.f, 0.f, 0.f, 0.f};                                        // S^T: register e <-> key k0 + 4 fq + e, query fr
#pragma unroll
    for (int s = 0; s < KS; ++s)
      st = __builtin_amdgcn_mfma_f32_16x16x4f32(sk[fr * LDK + 4 * s + fq], qf[s], st, 0, 0, 0);
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = k0 + 4 * fq + e;
      int visible = key < nk ? 1 : 0;
      if (MASK) {                                                          // branch-free: one clamped load + integer logic
        const int mk = kmask ? kmask[(long)b * nk + (key < nk ? key : nk - 1)] : 1;
        visible &= (mk != 0) & ((causal == 0) | (key <= qi));
      }
      st[e] = visible ? st[e] : -INFINITY;
      tmax = fmaxf(tmax, st[e]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(mrun, tmax);
    const float mref = (MASK && mnew == -INFINITY) ? 0.f : mnew;            // nothing visible yet: every p below is exp(-inf) = 0
    const float alpha = expf(mrun - mref);                                  // first tile: exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { st[e] = expf(st[e] - mref); psum += st[e]; }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    lrun = lrun * alpha + psum;
    mrun = mnew;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      o[d] *= alpha;                                                        // (same factor in all four registers: one query per lane)
#pragma unroll
      for (int s = 0; s < 4; ++s)                                           // A[i = dim 16d + fr][k] = V[key 4 fq + s][dim]
        o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[(4 * fq + s) * LDV + d * 16 + fr], st[s], o[d], 0, 0, 0);
    }
  }
  // O^T accumulator: column = query fr, row = dim 16d + 4 fq + e
  if (!q_ok) return;
  const float inv = 1.0f / lrun;
  float* op = out + ((long)b * nq + qi) * ldo + h * HD;
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int dim = d * 16 + 4 * fq + e;
      if (dim < HD) op[dim] = o[d][e] * inv;
    }
}

// CLIPTextEmbeddings in f32: out[b, t, :] = token_embedding[ids[b, t]] + position_embedding[t]
__global__ void __launch_bounds__(256)
k_clip_embed_f32(const int* __restrict__ ids, const float* __restrict__ tok, const float* __restrict__ pos, long rows, int n, int C,
                 int vocab, float* __restrict__ out) {
  const int c4 = C / 4;
  const long total = rows * c4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / c4; const int c = (int)(e - r * c4) * 4;
    int id = ids[r]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);       // out-of-range ids are clamped, not faulted on
    const f32x4 a = *reinterpret_cast<const f32x4*>(tok + (long)id * C + c);
    const f32x4 p = *reinterpret_cast<const f32x4*>(pos + (long)(r % n) * C + c);
    *reinterpret_cast<f32x4*>(out + r * C + c) = a + p;
  }
}

// conv_in, weights in LDS: a workgroup keeps the whole [9 cin][cout] weight (transposed: a thread's 4 output channels are one
// 16-byte LDS read per tap, conflict-free) and walks `ppb` pixels; thread = (pixel slot, quad of output channels); the 9 cin
// latent values of a pixel are wave-broadcast global loads.  Same products in the same (tap, channel) order as k_conv_in_f32
// below, which stays for shapes this form does not take (cout % 4, weights over 64 KB).
__global__ void __launch_bounds__(256)
k_conv_in_f32_lds(const float* __restrict__ lat, const float* __restrict__ w, const float* __restrict__ bias, int B, int cin, int H,
                  int W, int cout, int ppb, float* __restrict__ out) {
  extern __shared__ float swt[];                                             // [9 cin][cout]
  const int K = 9 * cin, cq = cout / 4;
  for (int e = threadIdx.x; e < K * cout; e += 256) { const int co = e / K, k = e - co * K; swt[k * cout + co] = w[e]; }
  __syncthreads();
  const int slots = 256 / cq;                                                // pixels in flight per pass
  const int slot = threadIdx.x / cq, q = threadIdx.x - slot * cq;
  if (slot >= slots) return;
  const long npix = (long)B * H * W;
  const long p0 = (long)blockIdx.x * ppb;
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * q);
  for (long pix = p0 + slot; pix < p0 + ppb && pix < npix; pix += slots) {
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    f32x4 acc = bv;
    for (int ty = 0; ty < 3; ++ty)
      for (int tx = 0; tx < 3; ++tx) {
        const int iy = y + ty - 1, ix = x + tx - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        for (int c = 0; c < cin; ++c) {
          const float v = lat[(((long)b * cin + c) * H + iy) * W + ix];
          const f32x4 wv = *reinterpret_cast<const f32x4*>(&swt[((ty * 3 + tx) * cin + c) * cout + 4 * q]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaf(v, wv[e], acc[e]);
        }
      }
    *reinterpret_cast<f32x4*>(out + pix * cout + 4 * q) = acc;
  }
}

// conv_in: 3x3, pad 1, fp32 NCHW latent -> NHWC f32; one thread per (pixel, output channel).  w is [Cout][ky][kx][Cin].
__global__ void __launch_bounds__(256)
k_conv_in_f32(const float* __restrict__ lat, const float* __restrict__ w, const float* __restrict__ bias, int B, int cin,
              int H, int W, int cout, float* __restrict__ out) {
  const long total = (long)B * H * W * cout;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int co = (int)(e % cout);
    const long pix = e / cout;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    float acc = bias[co];
    for (int ty = 0; ty < 3; ++ty)
      for (int tx = 0; tx < 3; ++tx) {
        const int iy = y + ty - 1, ix = x + tx - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        for (int c = 0; c < cin; ++c)
          acc = fmaf(lat[(((long)b * cin + c) * H + iy) * W + ix], w[((long)co * 9 + ty * 3 + tx) * cin + c], acc);
      }
    out[e] = acc;
  }
}

__global__ void k_temb_f32(float t_val, const float* __restrict__ t_dev, int B, int dim, float* __restrict__ out) {
  const float t = t_dev ? *t_dev : t_val;
  const int half = dim / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * dim; i += gridDim.x * blockDim.x) {
    const int kk = i % dim;
    const int f_i = kk < half ? kk : kk - half;
    const float f = expf(-9.210340371976184f * (float)f_i / (float)half);    // ln(10000)
    const float a = t * f;
    out[i] = kk < half ? cosf(a) : sinf(a);
  }
}

// ================================================================================================
// bf16x3 contractions on fp32 storage (dtype 3 of sdn_unet_config): every operand element x is split on the fly into
// hi = bf16(x) and lo = bf16(x - hi) (x = hi + lo to 2^-17 relative: 16 mantissa bits) and every product a.w is formed as
// a_hi w_hi + a_hi w_lo + a_lo w_hi on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, f32 accumulation; the dropped
// lo.lo term is below 2^-17 relative).  3 MFMAs at the 16-bit rate instead of 8 at the f32-input rate for the same
// 16 x 16 x 32 block: the precision mode's contraction at ~5x the f32-MFMA throughput, with everything else (storage,
// norms, softmax, epilogues) unchanged f32.  Measured distance from the pure-fp32 reference: ~1.5e-5 per UNet forward,
// 3.4e-5 over the 10-step loop in the emulation (profiles/round3_precision_ablation.md), 5.3e-5 / 5.5e-5 over 10 / 50 steps in the
// engine (profiles/round3_parity.json), against the north star's 1e-3.
// ================================================================================================
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 v2;
  v2 p = {(__bf16)a, (__bf16)b};                                             // v_cvt_pk_bf16_f32 (RNE)
  return *reinterpret_cast<unsigned*>(&p);
}
// two f32 -> packed hi pair, packed lo pair (x - float(hi) is exact in f32)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pk_bf16(a, b);
  lo = pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, u32x4& hi, u32x4& lo) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  split2(v0[0], v0[1], h0, l0); split2(v0[2], v0[3], h1, l1);
  split2(v1[0], v1[1], h2, l2); split2(v1[2], v1[3], h3, l3);
  hi = (u32x4){h0, h1, h2, h3}; lo = (u32x4){l0, l1, l2, l3};
}
__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
// acc += (ah + al) . (bh + bl) without the lo.lo term; small terms first
__device__ __forceinline__ f32x4 mfma_x3(const u32x4& ah, const u32x4& al, const u32x4& bh, const u32x4& bl, f32x4 c) {
  c = mfma_bf16(al, bh, c);
  c = mfma_bf16(ah, bl, c);
  return mfma_bf16(ah, bh, c);
}

// GEMM / implicit conv, same operator as k_gemm_f32.  128 x 160 tile (160 divides every SD-v1.4 width: 320, 640, 960, 1280,
// 2560), BK = 32, 4 waves (2 x 2), each 64 rows x 80 columns = 4 x 5 accumulators in the swapped orientation (weights = A
// operand).  Staging: thread -> (k chunk tid & 7 = 4 consecutive
// k, rows tid / 8 + 32 i): a wave-instruction fetches 8 whole 128-byte row segments; the four (activations) + five (weights) float4 are
// requested TWO k-tiles ahead into one of two register sets, and the tile one ahead is split into hi / lo and written into
// the other LDS stage in instalments placed between the MFMA groups of the current tile (the split is ~100 VALU
// operations per thread per k-tile: it has to run under the matrix pipe, not beside it).  LDS: 2 stages x (two images
// [128][32] + two [160][32]) bf16 with unpadded 64-byte rows = 72 KB (two workgroups per CU); 16-byte chunk c of row r sits at chunk
// c ^ (3 * ((r >> 2) & 1)): conflict-free for the four lane groups of ds_read_b128 on gfx950 (brute-forced).  One barrier
// per k-tile.  The three MFMAs of one accumulator are issued four MFMAs apart (term-major order).
constexpr int XM = 128, XK = 32;

// Every global load of the main loop is UNCONDITIONAL and branch-free (row indices clamped into the matrix: rows past M / N
// only feed outputs that are never stored; conv halo taps are fetched from a clamped pixel and zeroed by a select; the tile
// index past the end re-reads the last tile), so the loop body is one basic block and the compiler's s_waitcnt vmcnt(N)
// leaves the newest tile's requests in flight when the split consumes the older one.
// XNJ = 5: the 160-column tile; XNJ = 4: 128 columns (GEGLU projections: their value / gate column pairs must not straddle
// the two wave columns, and their widths are multiples of 128).
template <bool CONV, int XNJ>
__global__ void __launch_bounds__(256, 2)
k_gemm_x3(const GemmArgsF g) {
  constexpr int XN = 32 * XNJ;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2][2 * (XM + XN) * XK];   // [stage][A hi | A lo | W hi | W lo]
  constexpr int IMG_AL = XM * XK, IMG_WH = 2 * XM * XK, IMG_WL = (2 * XM + XN) * XK;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;
  const int tm = blockIdx.x / g.tiles_n, tn = blockIdx.x - tm * g.tiles_n;
  const int m0 = tm * XM, n0 = tn * XN;
  const int sc = tid & 7, sr = tid >> 3;                                     // staging role: k chunk, first row
  int cy[4], cx[4];
  long abase[4], wbase[XNJ];                                                 // element offset of (row, k chunk) in a / w
#pragma unroll
  for (int i = 0; i < XNJ; ++i) wbase[i] = (long)min(n0 + sr + 32 * i, g.N - 1) * g.K + 4 * sc;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int am = min(m0 + sr + 32 * i, g.M - 1);
    if (CONV) {
      const int hw = g.Ho * g.Wo;
      const int pb = am / hw;
      const int p = am - pb * hw;
      const int oy = p / g.Wo, ox = p - oy * g.Wo;
      cy[i] = oy * g.stride + g.conv_off; cx[i] = ox * g.stride + g.conv_off;
      abase[i] = (long)pb * g.Hs * g.Ws * g.Cin + 4 * sc;
    } else {
      abase[i] = (long)am;
    }
  }
  const int Hi = g.upsample ? 2 * g.Hs : g.Hs, Wi = g.upsample ? 2 * g.Ws : g.Ws;
  const int ld1 = g.K1, ld2 = g.K - g.K1;
  const int nkt = g.K / XK;

  auto load_a = [&](int kt, f32x4 (&v)[4]) {
    const int k0 = min(kt, nkt - 1) * XK;
    if (CONV) {
      const int tap = k0 / g.Cin, c0 = k0 - tap * g.Cin;                     // Cin % 64 == 0: a 32-wide k-tile lies inside one tap
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int y = cy[i] + dy, x = cx[i] + dx;
        const bool in = y >= 0 && y < Hi && x >= 0 && x < Wi;
        const int yc = min(max(y, 0), Hi - 1), xc = min(max(x, 0), Wi - 1);
        const int sy = g.upsample ? yc >> 1 : yc, sx = g.upsample ? xc >> 1 : xc;
        const f32x4 t = *reinterpret_cast<const f32x4*>(g.a + abase[i] + ((long)sy * g.Ws + sx) * g.Cin + c0);
        v[i] = in ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    } else {
      const bool first = k0 < g.K1;                                          // wave-uniform: which of the two sources
      const float* src = first ? g.a : g.a2;
      const long ld = first ? ld1 : ld2;
      const int kk = (first ? k0 : k0 - g.K1) + 4 * sc;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + abase[i] * ld + kk);
    }
  };
  auto load_w = [&](int kt, f32x4 (&v)[XNJ]) {
    const int k0 = min(kt, nkt - 1) * XK;
#pragma unroll
    for (int i = 0; i < XNJ; ++i) v[i] = *reinterpret_cast<const f32x4*>(g.w + wbase[i] + k0);
  };
  // LDS element offset of (row r, k chunk of 8 elements q) under the swizzle
  auto at = [](int r, int q) { return r * XK + ((q ^ (3 * ((r >> 2) & 1))) << 3); };
  // one instalment of the split: one float4 (4 consecutive k of row sr + 32 i) -> the hi and lo images of `stage`
#ifdef SDN_X3_ABLATE                       // timing-only ablations (tools/bench_x3_gemm.py): compiled in only on request, the runtime
  const int dbg = g.dbg;                   // branches cost the production kernel 13 % (measured: 2.78 -> 2.41 images/sec in the loop)
#else
  constexpr int dbg = 0;
#endif
  auto put = [&](int stage, int img_hi, int img_lo, const f32x4& v, int i) {
    unsigned h0, l0, h1, l1;
    if (dbg & 4) return;
    if (dbg & 1) { h0 = __float_as_uint(v[0]); l0 = __float_as_uint(v[1]); h1 = __float_as_uint(v[2]); l1 = __float_as_uint(v[3]); }
    else { split2(v[0], v[1], h0, l0); split2(v[2], v[3], h1, l1); }
    const int o = at(sr + 32 * i, sc >> 1) + 4 * (sc & 1);
    *reinterpret_cast<uint2*>(&smem[stage][img_hi + o]) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(&smem[stage][img_lo + o]) = make_uint2(l0, l1);
  };

  f32x4 acc[4][XNJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < XNJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 ra0[4], rw0[XNJ], ra1[4], rw1[XNJ];                                  // register set t & 1 carries k-tile t
  load_a(0, ra0); load_w(0, rw0);
  load_a(1, ra1); load_w(1, rw1);
#pragma unroll
  for (int i = 0; i < 4; ++i) put(0, 0, IMG_AL, ra0[i], i);
#pragma unroll
  for (int i = 0; i < XNJ; ++i) put(0, IMG_WH, IMG_WL, rw0[i], i);
  __syncthreads();

  // k-tile kt from LDS stage CUR; meanwhile tile kt + 2 is requested into the register set tile kt has vacated and tile
  // kt + 1 (register set NXT, requested one iteration ago) is split into stage NXT between the MFMA groups (past the last
  // tile both are harmless repeats of the last tile into a stage nobody reads)
  auto step = [&](auto CURc, int kt, f32x4 (&rac)[4], f32x4 (&rwc)[XNJ], f32x4 (&ran)[4], f32x4 (&rwn)[XNJ]) {
    constexpr int CUR = decltype(CURc)::value, NXT = CUR ^ 1;
    if (!(dbg & 2)) { load_a(kt + 2, rac); load_w(kt + 2, rwc); }
    u32x4 ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = at(wm * 64 + i * 16 + fr, fq);
      ah[i] = *reinterpret_cast<const u32x4*>(&smem[CUR][o]); al[i] = *reinterpret_cast<const u32x4*>(&smem[CUR][IMG_AL + o]);
    }
#pragma unroll
    for (int j = 0; j < XNJ; ++j) {
      const int o = at(wn * (XN / 2) + j * 16 + fr, fq);
      const u32x4 wh = *reinterpret_cast<const u32x4*>(&smem[CUR][IMG_WH + o]), wl = *reinterpret_cast<const u32x4*>(&smem[CUR][IMG_WL + o]);
      if (!(dbg & 8)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = mfma_bf16(wl, ah[i], acc[i][j]);
      }
      if (j < 4) put(NXT, 0, IMG_AL, ran[j], j);
      if (!(dbg & 8)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = mfma_bf16(wh, al[i], acc[i][j]);
      }
      put(NXT, IMG_WH, IMG_WL, rwn[j], j);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = mfma_bf16(wh, ah[i], acc[i][j]);
    }
    __syncthreads();
  };
  for (int kt = 0; kt < nkt; kt += 2) {
    step(std::integral_constant<int, 0>{}, kt, ra0, rw0, ra1, rw1);
    if (kt + 1 < nkt) step(std::integral_constant<int, 1>{}, kt + 1, ra1, rw1, ra0, rw0);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) gemm_row_epilogue<XNJ>(g, acc[i], m0 + wm * 64 + i * 16 + fr, n0 + wn * (XN / 2), fq);
}

// Attention, flash form, bf16x3 products.  Workgroup = 4 waves = 64 QS queries of one (batch, head), wave w owning QS sets of 16
// (query = block + 64 s + 16 w + lane % 16): every K / V fragment read from LDS feeds QS MFMAs and a staged tile serves
// 64 QS queries (the f32 K / V tiles are twice the bytes of the 16-bit kernels': with 64 queries per workgroup the launch
// was bound by re-fetching them).  Keys / values arrive in tiles of 32 (prefetched one tile ahead into registers, split into
// hi / lo on their way into LDS).  Per query set:
//   S^T[key][query] = K . Q^T per 16-key half tile: A = K fragment (8 consecutive dims of one key), B = Q fragment (scaled,
//     split once per kernel); a lane ends with 4 keys' scores of ONE query -- the softmax bookkeeping of k_attention_f32;
//   O^T[dim][query] += V^T . P^T as ONE 32-key contraction per 16-dim block: k slot (lane group g, s) <-> key 4 g + s (s < 4)
//     or 16 + 4 g + (s - 4): exactly the eight probabilities the lane already holds, so P goes from the score accumulators to
//     the B operand in registers; V^T sits in LDS [dim][32 key slots] in that slot order (transposed and permuted by staging).
template <int HD, int QS>
__global__ void __launch_bounds__(256)
k_attention_x3(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ out,
               int heads, int nq, int nk, int ldq, int ldk, int ldv, int ldo, float scale, int triple) {
  constexpr int NB = (HD + 31) / 32, HDP = NB * 32;                         // score product: head dim in blocks of 32
  constexpr int DB = (HD + 15) / 16, DP = DB * 16;                          // output dims in blocks of 16
  constexpr int LK = HDP + 8, LV = 40;                                      // LDS row lengths (bf16 elements)
  constexpr int NV4 = 32 * (HD / 4), NI = (NV4 + 255) / 256;                // float4 pieces of one K (or V) tile per thread
  constexpr int QB = 64 * QS;
  __shared__ __attribute__((aligned(16))) unsigned short sKh[32 * LK], sKl[32 * LK], sVh[DP * LV], sVl[DP * LV];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int qblocks = (nq + QB - 1) / QB;
  const int bh = blockIdx.x / qblocks, qb = blockIdx.x - bh * qblocks;
  const int b = bh / heads, h = bh - b * heads;
  // scores are kept in log2 units (log2 e folded into the Q scale): the exponentials are single v_exp_f32 instructions (the
  // library expf costs ~25 VALU operations; with 9 of them per 32 keys per query the kernel was bound by vector issue)
  const float scale2 = scale * 1.4426950408889634f;
  u32x4 qh[QS][NB], ql[QS][NB];                                             // B operand of block n: Q[query][32 n + 8 fq .. + 7] * scale
  f32x4 o[QS][DB];
  float mrun[QS], lrun[QS];
#pragma unroll
  for (int s = 0; s < QS; ++s) {
    const int qi = qb * QB + 64 * s + wid * 16 + fr;
    const float* qp = q + ((long)b * nq + (qi < nq ? qi : 0)) * ldq + h * HD;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int d0 = 32 * n + 8 * fq;
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = a;
      if (qi < nq && d0 < HD) { a = *reinterpret_cast<const f32x4*>(qp + d0) * scale2; c = *reinterpret_cast<const f32x4*>(qp + d0 + 4) * scale2; }
      split8(a, c, qh[s][n], ql[s][n]);
    }
#pragma unroll
    for (int d = 0; d < DB; ++d) o[s][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mrun[s] = -INFINITY; lrun[s] = 0.f;
  }
  for (int i = tid; i < 32 * LK; i += 256) { sKh[i] = 0; sKl[i] = 0; }      // padding dims HD..HDP stay zero
  for (int i = tid; i < DP * LV; i += 256) { sVh[i] = 0; sVl[i] = 0; }      // padding dims HD..DP stay zero
  const float* kb = k + (long)b * nk * ldk + h * HD;
  const float* vb = v + (long)b * nk * ldv + h * HD;
  f32x4 rk[NI], rv[NI];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int i = tid + it * 256;
      rk[it] = (f32x4){0.f, 0.f, 0.f, 0.f}; rv[it] = rk[it];
      if (i < NV4) {
        const int r = i / (HD / 4), c4 = (i - r * (HD / 4)) * 4;
        if (k0 + r < nk) {
          rk[it] = *reinterpret_cast<const f32x4*>(kb + (long)(k0 + r) * ldk + c4);
          rv[it] = *reinterpret_cast<const f32x4*>(vb + (long)(k0 + r) * ldv + c4);
        }
      }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < nk; k0 += 32) {
    __syncthreads();                                                       // previous tile fully consumed (and the zero fill done)
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int i = tid + it * 256;
      if (i < NV4) {
        const int r = i / (HD / 4), c4 = (i - r * (HD / 4)) * 4;
        unsigned h0, l0, h1, l1;
        split2(rk[it][0], rk[it][1], h0, l0); split2(rk[it][2], rk[it][3], h1, l1);
        *reinterpret_cast<uint2*>(&sKh[r * LK + c4]) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(&sKl[r * LK + c4]) = make_uint2(l0, l1);
        split2(rv[it][0], rv[it][1], h0, l0); split2(rv[it][2], rv[it][3], h1, l1);
        const int slot = 8 * ((r & 15) >> 2) + 4 * (r >> 4) + (r & 3);     // key 16 t + 4 g + e -> slot 8 g + 4 t + e
        sVh[(c4 + 0) * LV + slot] = (unsigned short)h0; sVh[(c4 + 1) * LV + slot] = (unsigned short)(h0 >> 16);
        sVh[(c4 + 2) * LV + slot] = (unsigned short)h1; sVh[(c4 + 3) * LV + slot] = (unsigned short)(h1 >> 16);
        sVl[(c4 + 0) * LV + slot] = (unsigned short)l0; sVl[(c4 + 1) * LV + slot] = (unsigned short)(l0 >> 16);
        sVl[(c4 + 2) * LV + slot] = (unsigned short)l1; sVl[(c4 + 3) * LV + slot] = (unsigned short)(l1 >> 16);
      }
    }
    __syncthreads();
    if (k0 + 32 < nk) fetch(k0 + 32);                                      // next tile in flight under the MFMAs
    f32x4 st[QS][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int s = 0; s < QS; ++s) st[s][t] = (f32x4){0.f, 0.f, 0.f, 0.f};  // register e <-> key k0 + 16 t + 4 fq + e, query fr
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const int off = (16 * t + fr) * LK + 32 * n + 8 * fq;
        const u32x4 kh = *reinterpret_cast<const u32x4*>(&sKh[off]), kl = *reinterpret_cast<const u32x4*>(&sKl[off]);
#pragma unroll
        for (int s = 0; s < QS; ++s) st[s][t] = mfma_bf16(kl, qh[s][n], st[s][t]);
#pragma unroll
        for (int s = 0; s < QS; ++s) st[s][t] = mfma_bf16(kh, ql[s][n], st[s][t]);
#pragma unroll
        for (int s = 0; s < QS; ++s) st[s][t] = mfma_bf16(kh, qh[s][n], st[s][t]);
      }
    }
    u32x4 ph[QS], pl[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
      if (k0 + 32 > nk) {                                                   // ragged last tile (wave-uniform)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k0 + 16 * t + 4 * fq + e >= nk) st[s][t][e] = -INFINITY;
      }
      float tmax = fmaxf(fmaxf(fmaxf(st[s][0][0], st[s][0][1]), fmaxf(st[s][0][2], st[s][0][3])),
                         fmaxf(fmaxf(st[s][1][0], st[s][1][1]), fmaxf(st[s][1][2], st[s][1][3])));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float mnew = fmaxf(mrun[s], tmax);
      const float alpha = __builtin_amdgcn_exp2f(mrun[s] - mnew);             // first tile: 2^(-inf) = 0
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { st[s][t][e] = __builtin_amdgcn_exp2f(st[s][t][e] - mnew); psum += st[s][t][e]; }
      psum += __shfl_xor(psum, 16, 64);
      psum += __shfl_xor(psum, 32, 64);
      lrun[s] = lrun[s] * alpha + psum;
      mrun[s] = mnew;
      split8(st[s][0], st[s][1], ph[s], pl[s]);                             // slots 8 fq + 0..3 = half tile 0, + 4..7 = half tile 1
#pragma unroll
      for (int d = 0; d < DB; ++d) o[s][d] *= alpha;
    }
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      const int off = (d * 16 + fr) * LV + 8 * fq;
      const u32x4 vh = *reinterpret_cast<const u32x4*>(&sVh[off]), vl = *reinterpret_cast<const u32x4*>(&sVl[off]);
#pragma unroll
      for (int s = 0; s < QS; ++s) o[s][d] = mfma_bf16(vl, ph[s], o[s][d]);
#pragma unroll
      for (int s = 0; s < QS; ++s) o[s][d] = mfma_bf16(vh, pl[s], o[s][d]);
#pragma unroll
      for (int s = 0; s < QS; ++s) o[s][d] = mfma_bf16(vh, ph[s], o[s][d]);
    }
  }
#pragma unroll
  for (int s = 0; s < QS; ++s) {
    const int qi = qb * QB + 64 * s + wid * 16 + fr;
    if (qi >= nq) continue;
    const float inv = 1.0f / lrun[s];
    if (triple) {                                                           // [hi(ldo) | lo(ldo) | hi(ldo)] rows: the to_out GEMM's A operand
      unsigned short* row3 = reinterpret_cast<unsigned short*>(out) + ((long)b * nq + qi) * 3 * ldo;
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const int dim = d * 16 + 4 * fq;                                    // HD % 4 == 0: a quad is valid or not as a whole
        if (dim < HD) {
          store_triple2(row3, ldo, h * HD + dim, o[s][d][0] * inv, o[s][d][1] * inv);
          store_triple2(row3, ldo, h * HD + dim + 2, o[s][d][2] * inv, o[s][d][3] * inv);
        }
      }
      continue;
    }
    float* op = out + ((long)b * nq + qi) * ldo + h * HD;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int dim = d * 16 + 4 * fq + e;
        if (dim < HD) op[dim] = o[s][d][e] * inv;
      }
  }
}

// f32 [rows, c1] (++ f32 [rows, c2]) -> triple [rows, 3 (c1 + c2)]: raw residual-stream tensors a GEMM reads (shortcut convs over
// the skip concatenation, down / up-sampling convs, proj_out), text states.  One pass: 4 B read, 6 B written per element.
__global__ void __launch_bounds__(256)
k_split3(const float* __restrict__ x, const float* __restrict__ x2, long rows, int c1, int c2, unsigned short* __restrict__ out) {
  const int C = c1 + c2, cq = C / 4;                                         // c1, c2 % 4 == 0 (host-checked)
  const long total = rows * cq;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / cq; const int c = (int)(e - r * cq) * 4;
    const f32x4 v = c < c1 ? *reinterpret_cast<const f32x4*>(x + r * c1 + c) : *reinterpret_cast<const f32x4*>(x2 + r * c2 + (c - c1));
    unsigned short* row3 = out + r * 3 * C;
    const unsigned h0 = bf16_rne(v[0]), h1 = bf16_rne(v[1]), h2 = bf16_rne(v[2]), h3 = bf16_rne(v[3]);
    const unsigned l0 = bf16_rne(v[0] - __uint_as_float(h0 << 16)), l1 = bf16_rne(v[1] - __uint_as_float(h1 << 16));
    const unsigned l2 = bf16_rne(v[2] - __uint_as_float(h2 << 16)), l3 = bf16_rne(v[3] - __uint_as_float(h3 << 16));
    const uint2 hi = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16)), lo = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
    *reinterpret_cast<uint2*>(row3 + c) = hi;
    *reinterpret_cast<uint2*>(row3 + C + c) = lo;
    *reinterpret_cast<uint2*>(row3 + 2 * C + c) = hi;
  }
}

// f32 W [rows, cols] -> bf16 [rows, 3 cols]: every K-group of `group` columns (cols itself, or the Cin of one tap of a
// [O][ky][kx][Cin] conv weight) becomes [hi(group) | hi(group) | lo(group)] -- the partner of the activations' [hi | lo | hi]
__global__ void __launch_bounds__(256)
k_expand3_weights(const float* __restrict__ w, long rows, int cols, int group, unsigned short* __restrict__ out) {
  const long total = rows * cols;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / cols; const int c = (int)(e - r * cols);
    const int gi = c / group, ci = c - gi * group;
    const float v = w[e];
    const unsigned short hi = bf16_rne(v), lo = bf16_rne(v - __uint_as_float((unsigned)hi << 16));
    unsigned short* o = out + r * 3 * cols + (long)gi * 3 * group + ci;
    o[0] = hi; o[group] = hi; o[2 * group] = lo;
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// x3 = 0: f32-input MFMA (k_gemm_f32); 1: bf16x3 split products (k_gemm_x3).  Same validation, same GemmArgsF.
static int gemm_f32_storage(int x3, const sdn_gemm_desc* d, const void* a, const void* a2, const void* w, const float* bias,
                            const float* rowbias, const float* rowgate, const void* residual, void* out, void* stream) {
  if (!d || !a || !w || !out) return SDN_E_INVALID;
  if (d->M < 0 || d->N <= 0 || d->K <= 0 || (d->K % 64) != 0 || (d->N % 32) != 0 || d->split_k > 1) return SDN_E_INVALID;
  if (d->M == 0) return SDN_OK;
  const int n_valid = d->n_valid > 0 ? d->n_valid : d->N;
  if (n_valid > d->N || d->act < 0 || d->act > 4 || d->out_kind < 0 || d->out_kind > 2) return SDN_E_INVALID;
  if (!al16(a) || !al16(w) || (a2 && !al16(a2)) || (bias && !al16(bias)) || (rowbias && !al16(rowbias)) ||
      (rowgate && !al16(rowgate)) || (reinterpret_cast<uintptr_t>(out) & 15))
    return SDN_E_INVALID;
  GemmArgsF g{};
  g.a = (const float*)a; g.a2 = (const float*)a2; g.w = (const float*)w; g.bias = bias; g.rowbias = rowbias;
  g.rowgate = rowgate; g.residual = (const float*)residual; g.out = (float*)out;
  g.M = d->M; g.N = d->N; g.K = d->K; g.a_mode = d->a_mode;
  if (d->a_mode == SDN_A_PLAIN) {
    g.K1 = (d->K1 > 0 && d->K1 < d->K) ? d->K1 : d->K;
    if (g.K1 != d->K && (!a2 || (g.K1 % 64) != 0)) return SDN_E_INVALID;
  } else if (d->a_mode == SDN_A_CONV3X3) {
    if (d->Cin <= 0 || (d->Cin % 64) != 0 || d->K != 9 * d->Cin || d->Hs <= 0 || d->Ws <= 0 || d->Ho <= 0 || d->Wo <= 0 ||
        (d->stride != 1 && d->stride != 2) || d->M % (d->Ho * d->Wo) != 0)
      return SDN_E_INVALID;
    const int Hi = d->upsample ? 2 * d->Hs : d->Hs, Wi = d->upsample ? 2 * d->Ws : d->Ws;
    if (d->asym_pad != 0 && (d->asym_pad != 1 || d->stride != 2 || d->upsample)) return SDN_E_INVALID;
    const int pad2 = d->asym_pad ? 1 : 2;
    if ((Hi + pad2 - 3) / d->stride + 1 != d->Ho || (Wi + pad2 - 3) / d->stride + 1 != d->Wo) return SDN_E_INVALID;
    g.K1 = d->K; g.Hs = d->Hs; g.Ws = d->Ws; g.Cin = d->Cin; g.Ho = d->Ho; g.Wo = d->Wo; g.stride = d->stride;
    g.upsample = d->upsample; g.conv_off = d->asym_pad ? 1 : 0;
  } else {
    return SDN_E_INVALID;
  }
  if (d->act == SDN_ACT_GEGLU && (d->out_kind != SDN_OUT_BF16 || rowbias || rowgate || residual || n_valid != d->N || (d->N % 64) != 0))
    return SDN_E_INVALID;
  if ((rowbias || rowgate || d->residual_bcast || d->out_kind == SDN_OUT_F32_NCHW) && d->rows_per_batch <= 0) return SDN_E_INVALID;
  g.act = d->act; g.out_kind = d->out_kind; g.rows_per_batch = d->rows_per_batch; g.ld_rowbias = d->ld_rowbias;
  g.ld_rowgate = d->ld_rowgate; g.residual_bcast = d->residual_bcast; g.n_valid = n_valid;
  g.ldc = d->ldc > 0 ? d->ldc : (d->act == SDN_ACT_GEGLU ? d->N / 2 : n_valid);
  if (d->act == SDN_ACT_GEGLU && (g.ldc & 3)) return SDN_E_INVALID;
  if (x3 && d->M >= 64) {                                                    // (a handful of rows: the small f32 tile is as fast and exact)
    const bool wide = d->act != SDN_ACT_GEGLU && (d->N % 160 == 0 || d->N % 128 != 0);
    const int xn = wide ? 160 : 128;
    static const int x3_dbg = getenv("SDN_X3_DBG") ? atoi(getenv("SDN_X3_DBG")) : 0;
    g.dbg = x3_dbg;
    g.tiles_n = (d->N + xn - 1) / xn;
    const long tiles = (long)((d->M + XM - 1) / XM) * g.tiles_n;
    if (tiles > 0x7fffffffL) return SDN_E_INVALID;
    const dim3 grid((unsigned)tiles), block(256);
    if (g.a_mode == 1) {
      if (wide) hipLaunchKernelGGL((k_gemm_x3<true, 5>), grid, block, 0, (hipStream_t)stream, g);
      else hipLaunchKernelGGL((k_gemm_x3<true, 4>), grid, block, 0, (hipStream_t)stream, g);
    } else {
      if (wide) hipLaunchKernelGGL((k_gemm_x3<false, 5>), grid, block, 0, (hipStream_t)stream, g);
      else hipLaunchKernelGGL((k_gemm_x3<false, 4>), grid, block, 0, (hipStream_t)stream, g);
    }
    return sdn_launch_status();
  }
  g.tiles_n = (d->N + BN - 1) / BN;
  const long tiles = (long)((d->M + BM - 1) / BM) * g.tiles_n;
  if (tiles > 0x7fffffffL) return SDN_E_INVALID;
  hipLaunchKernelGGL(k_gemm_f32, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, g);
  return sdn_launch_status();
}

extern "C" int sdn_gemm_f32(const sdn_gemm_desc* d, const void* a, const void* a2, const void* w, const float* bias,
                            const float* rowbias, const float* rowgate, const void* residual, void* out, void* stream) {
  return gemm_f32_storage(0, d, a, a2, w, bias, rowbias, rowgate, residual, out, stream);
}
extern "C" int sdn_gemm_x3(const sdn_gemm_desc* d, const void* a, const void* a2, const void* w, const float* bias,
                           const float* rowbias, const float* rowgate, const void* residual, void* out, void* stream) {
  return gemm_f32_storage(1, d, a, a2, w, bias, rowbias, rowgate, residual, out, stream);
}

// SDN_GN_F32_OLD=1 selects the per-group kernels (A/B switch): looked up once, not on each of the ~60 launches of a forward
static bool gn_f32_old_form() {
  static const bool v = getenv("SDN_GN_F32_OLD") != nullptr;
  return v;
}

static int groupnorm_f32_impl(int triple, const void* x, const void* x2, int32_t batch, int32_t hw, int32_t c1, int32_t c2,
                              int32_t groups, float eps, int32_t silu, const float* gamma, const float* beta, void* out,
                              float* stats_ws, void* stream) {
  (void)stats_ws;                                                           // the 16-bit kernels' scratch: not needed here
  if (!x || !gamma || !beta || !out || batch < 0 || hw <= 0 || c1 <= 0 || c2 < 0 || groups <= 0 || (c2 > 0 && !x2) ||
      (c1 + c2) % groups != 0)
    return SDN_E_INVALID;
  if (batch == 0) return SDN_OK;
  const int cpg = (c1 + c2) / groups;
  // row-major two-pass form (coalesced rows; chunk partials in the caller's scratch: B * 129 * groups * 2 floats cover 64 chunks)
  if (stats_ws && (c1 & 3) == 0 && (c2 & 3) == 0 && c1 + c2 <= GN_MAXC && groups <= 64 && al16(x) && (!x2 || al16(x2)) && al16(gamma) &&
      al16(beta) && (reinterpret_cast<uintptr_t>(stats_ws) & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & (triple ? 7 : 15)) == 0 &&
      !gn_f32_old_form()) {
    int rpc = 16;                                                            // rows per chunk: >= 16, at most 64 chunks per sample
    while ((hw + rpc - 1) / rpc > 64) rpc *= 2;
    const int nchunk = (hw + rpc - 1) / rpc;
    hipLaunchKernelGGL(k_gn_rows_stats, dim3((unsigned)(batch * nchunk)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (const float*)x2, hw, c1, c2, groups, rpc, (double*)stats_ws);
    hipLaunchKernelGGL(k_gn_rows_apply, dim3((unsigned)(batch * nchunk)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (const float*)x2, hw, c1, c2, groups, eps, silu, gamma, beta, rpc, (const double*)stats_ws, (float*)out, triple);
    return sdn_launch_status();
  }
  const bool pairs = (cpg & 1) == 0 && (c1 & 1) == 0 && cpg <= 512 && (reinterpret_cast<uintptr_t>(x) & 7) == 0 &&
                     (reinterpret_cast<uintptr_t>(out) & 7) == 0 && (!x2 || (reinterpret_cast<uintptr_t>(x2) & 7) == 0);
  if (pairs)
    hipLaunchKernelGGL(k_groupnorm_f32, dim3((unsigned)(batch * groups)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (const float*)x2, hw, c1, c2, groups, eps, silu, gamma, beta, (float*)out, triple);
  else
    hipLaunchKernelGGL(k_groupnorm_f32_any, dim3((unsigned)(batch * groups)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (const float*)x2, hw, c1, c2, groups, eps, silu, gamma, beta, (float*)out, triple);
  return sdn_launch_status();
}
extern "C" int sdn_groupnorm_f32(const void* x, const void* x2, int32_t batch, int32_t hw, int32_t c1, int32_t c2,
                                 int32_t groups, float eps, int32_t silu, const float* gamma, const float* beta, void* out,
                                 float* stats_ws, void* stream) {
  return groupnorm_f32_impl(0, x, x2, batch, hw, c1, c2, groups, eps, silu, gamma, beta, out, stats_ws, stream);
}
extern "C" int sdn_groupnorm_f32_triple(const void* x, const void* x2, int32_t batch, int32_t hw, int32_t c1, int32_t c2,
                                        int32_t groups, float eps, int32_t silu, const float* gamma, const float* beta,
                                        void* out_triple, float* stats_ws, void* stream) {
  return groupnorm_f32_impl(1, x, x2, batch, hw, c1, c2, groups, eps, silu, gamma, beta, out_triple, stats_ws, stream);
}

// LayerNorm with the row held in registers (C % 4 == 0, C <= 64 * 4 * LN_MAXV): one 16-byte load per lane and quad, the mean and
// the centred sum of squares from the registers, one 16-byte f32 store or three 8-byte triple stores.  One wave per row.
constexpr int LN_MAXV = 5;                                                   // C <= 1280
__global__ void __launch_bounds__(256)
k_layernorm_f32_regs(const float* __restrict__ x, long rows, int c, float eps, const float* __restrict__ gamma,
                     const float* __restrict__ beta, float* __restrict__ out, int triple) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, c4 = c >> 2;
  const float* xr = x + row * c;
  f32x4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int q = lane + 64 * j;
    v[j] = q < c4 ? *reinterpret_cast<const f32x4*>(xr + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
    s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  }
  const float mean = wave_sum(s) / (float)c;
  float qs = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    if (lane + 64 * j < c4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mean; qs = fmaf(d, d, qs); }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(qs) / (float)c + eps);
  unsigned short* row3 = reinterpret_cast<unsigned short*>(out) + row * 3 * c;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int q = lane + 64 * j;
    if (q >= c4) continue;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + 4 * q), be = *reinterpret_cast<const f32x4*>(beta + 4 * q);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * ga[e] + be[e];
    if (triple) {
      store_triple2(row3, c, 4 * q, o[0], o[1]);
      store_triple2(row3, c, 4 * q + 2, o[2], o[3]);
    } else {
      *reinterpret_cast<f32x4*>(out + row * c + 4 * q) = o;
    }
  }
}

static int layernorm_f32_impl(int triple, const void* x, int64_t rows, int32_t c, float eps, const float* gamma, const float* beta,
                              void* out, void* stream) {
  if (!x || !gamma || !beta || !out || rows < 0 || c <= 0 || (triple && (reinterpret_cast<uintptr_t>(out) & 3))) return SDN_E_INVALID;
  if (rows == 0) return SDN_OK;
  if ((c & 3) == 0 && c <= 64 * 4 * LN_MAXV && al16(x) && al16(gamma) && al16(beta) && (reinterpret_cast<uintptr_t>(out) & (triple ? 7 : 15)) == 0)
    hipLaunchKernelGGL(k_layernorm_f32_regs, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (long)rows, c, eps, gamma, beta, (float*)out, triple);
  else
    hipLaunchKernelGGL(k_layernorm_f32, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (long)rows, c, eps, gamma, beta, (float*)out, triple);
  return sdn_launch_status();
}
extern "C" int sdn_layernorm_f32(const void* x, int64_t rows, int32_t c, float eps, const float* gamma, const float* beta,
                                 void* out, void* stream) {
  return layernorm_f32_impl(0, x, rows, c, eps, gamma, beta, out, stream);
}
extern "C" int sdn_layernorm_f32_triple(const void* x, int64_t rows, int32_t c, float eps, const float* gamma, const float* beta,
                                        void* out_triple, void* stream) {
  return layernorm_f32_impl(1, x, rows, c, eps, gamma, beta, out_triple, stream);
}

static int attention_f32_storage(int x3, int triple, const void* q, const void* k, const void* v, void* out, int32_t batch, int32_t heads,
                                 int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                 float scale, void* stream) {
  if (!q || !k || !v || !out || batch < 0 || heads <= 0 || nq <= 0 || nk <= 0) return SDN_E_INVALID;
  if (!al16(k) || !al16(v) || (ldk & 3) || (ldv & 3)) return SDN_E_INVALID;
  if (x3 && (!al16(q) || (ldq & 3))) return SDN_E_INVALID;                   // the split kernel fetches Q as float4
  if (triple && (!x3 || (ldo & 1) || (reinterpret_cast<uintptr_t>(out) & 3))) return SDN_E_INVALID;
  if (batch == 0) return SDN_OK;
  const long grid = (long)batch * heads * ((nq + 63) / 64);
  if (grid > 0x7fffffffL) return SDN_E_INVALID;
  const float* qf = (const float*)q; const float* kf = (const float*)k; const float* vf = (const float*)v;
  // query sets per wave of the split kernel: as many as the registers hold, but not more than the query count fills
  static const int qs_cap = getenv("SDN_X3_QS") ? atoi(getenv("SDN_X3_QS")) : 2;      /* tuning knob (tools/bench_precision.py) */
#define SDN_ATTN_F32(HD, QSMAX)                                                                                       \
  if (x3) {                                                                                                           \
    const int QS_ = QSMAX < qs_cap ? QSMAX : qs_cap;                                                                                        \
    if (QS_ >= 4 && nq >= 256)                                                                                        \
      hipLaunchKernelGGL((k_attention_x3<HD, (QSMAX >= 4 ? 4 : 1)>), dim3((unsigned)((long)batch * heads * ((nq + 255) / 256))),  \
                         dim3(256), 0, (hipStream_t)stream, qf, kf, vf, (float*)out, heads, nq, nk, ldq, ldk, ldv, ldo, scale, triple); \
    else if (QS_ >= 2 && nq >= 128)                                                                                   \
      hipLaunchKernelGGL((k_attention_x3<HD, (QSMAX >= 2 ? 2 : 1)>), dim3((unsigned)((long)batch * heads * ((nq + 127) / 128))),  \
                         dim3(256), 0, (hipStream_t)stream, qf, kf, vf, (float*)out, heads, nq, nk, ldq, ldk, ldv, ldo, scale, triple); \
    else                                                                                                              \
      hipLaunchKernelGGL((k_attention_x3<HD, 1>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, qf, kf, vf, \
                         (float*)out, heads, nq, nk, ldq, ldk, ldv, ldo, scale, triple);                              \
  } else hipLaunchKernelGGL((k_attention_f32<HD>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, qf, kf, vf, \
                            (float*)out, heads, nq, nk, ldq, ldk, ldv, ldo, scale)
  switch (head_dim) {
    case 40: SDN_ATTN_F32(40, 4); break;
    case 64: SDN_ATTN_F32(64, 4); break;
    case 80: SDN_ATTN_F32(80, 2); break;
    case 160: SDN_ATTN_F32(160, 1); break;
    default: return SDN_E_INVALID;
  }
#undef SDN_ATTN_F32
  return sdn_launch_status();
}

extern "C" int sdn_attention_f32(const void* q, const void* k, const void* v, void* out, int32_t batch, int32_t heads,
                                 int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                 float scale, void* stream) {
  return attention_f32_storage(0, 0, q, k, v, out, batch, heads, nq, nk, head_dim, ldq, ldk, ldv, ldo, scale, stream);
}
extern "C" int sdn_attention_x3(const void* q, const void* k, const void* v, void* out, int32_t batch, int32_t heads,
                                int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                float scale, void* stream) {
  return attention_f32_storage(1, 0, q, k, v, out, batch, heads, nq, nk, head_dim, ldq, ldk, ldv, ldo, scale, stream);
}
extern "C" int sdn_attention_x3_triple(const void* q, const void* k, const void* v, void* out_triple, int32_t batch, int32_t heads,
                                       int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                       float scale, void* stream) {
  return attention_f32_storage(1, 1, q, k, v, out_triple, batch, heads, nq, nk, head_dim, ldq, ldk, ldv, ldo, scale, stream);
}

extern "C" int sdn_split3(const float* x, const float* x2, int64_t rows, int32_t c1, int32_t c2, void* out_triple, void* stream) {
  if (!x || !out_triple || rows < 0 || c1 <= 0 || c2 < 0 || (c1 & 3) || (c2 & 3) || (c2 > 0 && !x2) || !al16(x) || (x2 && !al16(x2)) ||
      (reinterpret_cast<uintptr_t>(out_triple) & 7))
    return SDN_E_INVALID;
  if (rows == 0) return SDN_OK;
  long g = (rows * ((c1 + c2) / 4) + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(k_split3, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, x2, (long)rows, c1, c2, (unsigned short*)out_triple);
  return sdn_launch_status();
}

extern "C" int sdn_expand3_weights(const float* w, int64_t rows, int32_t cols, int32_t group, void* out_bf16, void* stream) {
  if (!w || !out_bf16 || rows < 0 || cols <= 0 || group <= 0 || cols % group != 0) return SDN_E_INVALID;
  if (rows == 0) return SDN_OK;
  long g = (rows * cols + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(k_expand3_weights, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, (long)rows, cols, group,
                     (unsigned short*)out_bf16);
  return sdn_launch_status();
}

extern "C" int sdn_conv_in_f32(const float* latents_nchw, const void* w, const float* bias, int32_t batch, int32_t cin,
                               int32_t h, int32_t wd, int32_t cout, void* out_nhwc, void* stream) {
  if (!latents_nchw || !w || !bias || !out_nhwc || batch < 0 || cin <= 0 || h <= 0 || wd <= 0 || cout <= 0) return SDN_E_INVALID;
  if (batch == 0) return SDN_OK;
  const size_t wbytes = (size_t)9 * cin * cout * 4;
  if ((cout & 3) == 0 && cout / 4 <= 256 && wbytes <= 64 * 1024 && (reinterpret_cast<uintptr_t>(out_nhwc) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(bias) & 15) == 0) {
    const int ppb = 64;
    const long npix = (long)batch * h * wd;
    hipLaunchKernelGGL(k_conv_in_f32_lds, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), wbytes, (hipStream_t)stream, latents_nchw,
                       (const float*)w, bias, batch, cin, h, wd, cout, ppb, (float*)out_nhwc);
    return sdn_launch_status();
  }
  const long total = (long)batch * h * wd * cout;
  long grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_conv_in_f32, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, latents_nchw, (const float*)w, bias,
                     batch, cin, h, wd, cout, (float*)out_nhwc);
  return sdn_launch_status();
}

int sdn_temb_f32(float timestep, const float* t_dev, int batch, int dim, void* out, void* stream) {
  if (!out || batch <= 0 || dim <= 0 || (dim & 1)) return SDN_E_INVALID;
  const int n = batch * dim;
  hipLaunchKernelGGL(k_temb_f32, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, timestep, t_dev, batch, dim, (float*)out);
  return sdn_launch_status();
}

extern "C" int sdn_timestep_embed_f32(float timestep, int32_t batch, int32_t dim, void* out, void* stream) {
  return sdn_temb_f32(timestep, nullptr, batch, dim, out, stream);
}

// CLIP text encoder in the fp32-storage modes (dtype 2 / 3 of sdn_clip_config): the causal (+ key padding) attention of its 12
// heads of 64 runs on the f32-input matrix cores in BOTH modes -- it is 1.7 % of the encoder's FLOPs (77 keys), and exact f32
// products keep the softmax's inputs at the reference's precision.
extern "C" int sdn_masked_attention_f32(const void* q, const void* k, const void* v, void* out, const int32_t* key_mask,
                                        int32_t causal, int32_t batch, int32_t heads, int32_t n, int32_t head_dim, int32_t ldq,
                                        int32_t ldk, int32_t ldv, int32_t ldo, float scale, void* stream) {
  if (!q || !k || !v || !out || batch < 0 || heads <= 0 || n <= 0 || head_dim != 64 || (!causal && !key_mask) || ldq < heads * 64 ||
      ldk < heads * 64 || ldv < heads * 64 || ldo < heads * 64 || ((ldk | ldv) & 3) ||
      ((reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15))
    return SDN_E_INVALID;
  if (batch == 0) return SDN_OK;
  const long blocks = (long)batch * heads * ((n + 63) / 64);
  if (blocks > 0x7fffffffL) return SDN_E_INVALID;
  hipLaunchKernelGGL((k_attention_f32<64, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)q,
                     (const float*)k, (const float*)v, (float*)out, heads, n, n, ldq, ldk, ldv, ldo, scale, causal ? 1 : 0, key_mask);
  return sdn_launch_status();
}

extern "C" int sdn_clip_embed_f32(const int32_t* input_ids, const void* token_embedding, const void* position_embedding, int64_t rows,
                                  int32_t seq_len, int32_t hidden, int32_t vocab, void* out, void* stream) {
  if (!input_ids || !token_embedding || !position_embedding || !out || rows < 0 || seq_len <= 0 || hidden <= 0 || (hidden & 3) ||
      vocab <= 0 || ((reinterpret_cast<uintptr_t>(token_embedding) | reinterpret_cast<uintptr_t>(position_embedding) |
                      reinterpret_cast<uintptr_t>(out)) & 15))
    return SDN_E_INVALID;
  if (rows == 0) return SDN_OK;
  long g = (rows * (hidden / 4) + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_clip_embed_f32, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, input_ids, (const float*)token_embedding,
                     (const float*)position_embedding, (long)rows, seq_len, hidden, vocab, (float*)out);
  return sdn_launch_status();
}
