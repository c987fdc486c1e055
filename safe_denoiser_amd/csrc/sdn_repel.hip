// Repellency projection (rows R1-R5): kernel-weighted mean of proj_ref rows + in-place axpy, batched over
// N queries, fp32.
//
// Data layout: proj_ref R is [M, D] fp32 row-major (D = C*H*W, the reference's NCHW flattening), 64 KiB per
// row for SD-v1.4.  Algorithmic traffic per call = ONE read of R (M*D*4 B) + x in/out.  The projection needs
// R twice (distances, then the weighted sum); M*D*4 = 33.75 MB (135 MB SD-v3) stays resident in the 256 MiB
// Infinity Cache between the two sweeps, so the second sweep does not return to HBM.
//
// Both sweeps are contractions over R -- N*M*D multiply-adds each, 1.08 GFLOP at N = 64, M = 515, D = 16384 -- so for a
// batch of prompts they are as much a matter of fp32 FLOPs as of bytes: both run on the f32-input matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32 products, f32 accumulation), each element of R fetched exactly ONCE per sweep by
// exactly one workgroup, 16 bytes per lane:
//   k_qnorm   (SD-v3 only) xq = x / ||x||_channel per pixel
//   k_gram    G[n,m] = sum_j xq[n,j] R[m,j] for a (64 refs x <=64 queries x column slice) block, + the slice's share of
//             |r_m|^2 and |x_n|^2; column slices give the grid its width (>= one workgroup per CU); per-slice partials
//   k_weights d2 = |x|^2 + |r|^2 - 2 G (slices summed in fixed order), w[n,m] (RBF or SPARSE), den[n], is_negation[n]
//   k_wsum    neg[n,j] = sum_m w[n,m] R[m,j] for a (<=64 queries x 64 columns) block, the four waves taking interleaved
//             reference rows and combining through LDS in fixed order; epilogue fused: neg / den, the in-place update of
//             x, the optional negative-score output.  For N <= 4 it derives the weights itself from the slice partials
//             (every workgroup redundantly: a few KB), so a single-prompt call -- the reference's shape -- is 2 launches.
// The Gram form's cancellation is harmless here: a query is never close to a reference in units of |x|^2 + |r|^2 (the
// reference's own torch.cdist takes the same form above 25 rows); fp32 error in d is ~1e-7 * (|x|^2 + |r|^2) / (2 d).
// Results are deterministic (no float atomics; every sum has a fixed order).
#include "sdn_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxSplits = 64;

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct Plan {
  int np;                 // N rounded up to 16 (rows of the padded per-query buffers)
  int mp;                 // M rounded up to 64
  int mquads;             // 64-reference blocks
  int splits, cps;        // column slices of the Gram sweep, columns per slice (multiple of 16)
  int ngroups;            // 64-query groups
  size_t off_xq, off_g, off_rr, off_xx, off_w, off_den, off_d2, total;
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

inline Plan make_plan(int N, int M, int64_t D) {
  Plan p;
  p.np = (N + 15) / 16 * 16; if (p.np < 16) p.np = 16;
  p.mp = (M + 63) / 64 * 64; if (p.mp < 64) p.mp = 64;
  p.mquads = p.mp / 64;
  p.ngroups = (N + 63) / 64; if (p.ngroups < 1) p.ngroups = 1;
  // enough column slices for ~1.1 workgroups per CU, slices of at least 64 columns
  int64_t want = (288 + (int64_t)p.mquads * p.ngroups - 1) / ((int64_t)p.mquads * p.ngroups);
  if (want < 1) want = 1;
  if (want > kMaxSplits) want = kMaxSplits;
  int64_t cps = (D + want - 1) / want;
  cps = (cps + 63) / 64 * 64;
  p.cps = (int)cps;
  p.splits = (int)((D + cps - 1) / cps);
  size_t o = 0;
  p.off_xq = o;  o += align256((size_t)N * D * 4);
  p.off_g = o;   o += align256((size_t)p.splits * p.np * p.mp * 4);
  p.off_rr = o;  o += align256((size_t)p.splits * p.mp * 4);
  p.off_xx = o;  o += align256((size_t)p.splits * p.np * 4);
  p.off_w = o;   o += align256((size_t)p.np * p.mp * 4);
  p.off_den = o; o += align256((size_t)p.np * 4);
  p.off_d2 = o;  o += align256((size_t)N * M * 4);
  p.total = o;
  return p;
}

// ---- channel-normalise the query (fast_sdv3:239) ------------------------------------------------
__global__ void __launch_bounds__(kThreads)
k_qnorm(const float* __restrict__ x, float* __restrict__ xq, int C, int HW) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= HW) return;
  const float* xn = x + (int64_t)n * C * HW;
  float s = 0.f;
  for (int c = 0; c < C; ++c) { float v = xn[(int64_t)c * HW + p]; s += v * v; }
  const float nrm = sqrtf(s);                    // 0 -> x/0 = NaN/Inf, as in the reference
  float* on = xq + (int64_t)n * C * HW;
  for (int c = 0; c < C; ++c) on[(int64_t)c * HW + p] = xn[(int64_t)c * HW + p] / nrm;
}

// ---- Gram sweep --------------------------------------------------------------------------------------------------
// Workgroup = (64-reference block, column slice, 64-query group); wave w owns references m0 = 64 q + 16 w .. +15.
// MFMA orientation: A[i = ref][k], B[k][j = query]; a lane (i | j = lane & 15, g = lane >> 4) loads 4 consecutive columns
// (16 B) of its reference row and of its query rows, and the four elements feed four consecutive MFMAs (MFMA e sees
// column c + 4 g + e at k = g: any assignment of columns to k slots is a valid contraction as long as A and B agree).
template <int NQB>
__global__ void __launch_bounds__(kThreads)
k_gram(const float* __restrict__ xq, const float* __restrict__ R, int N, int M, int64_t D, int cps, int np, int mp,
       float* __restrict__ gpart, float* __restrict__ rrpart, float* __restrict__ xxpart) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int mq = blockIdx.x, sp = blockIdx.y, ng = blockIdx.z;
  const int m0 = mq * 64 + wid * 16, n0 = ng * 64;
  const int64_t c_lo = (int64_t)sp * cps;
  const int64_t c_hi = c_lo + cps < D ? c_lo + cps : D;
  const float* rrow = R + (int64_t)(m0 + li < M ? m0 + li : (M > 0 ? M - 1 : 0)) * D;
  const float* xrow[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const int n = n0 + qb * 16 + li;
    xrow[qb] = xq + (int64_t)(n < N ? n : N - 1) * D;
  }
  f32x4 acc[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) acc[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float rr = 0.f, xx[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) xx[qb] = 0.f;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int64_t c = c_lo; c < c_hi; c += 16) {
    const int64_t col = c + 4 * g;
    const bool ok = col < c_hi;                                     // D % 4 == 0: a float4 is inside or outside as a whole
    const f32x4 rv = ok ? *reinterpret_cast<const f32x4*>(rrow + col) : zero4;
    rr = fmaf(rv[0], rv[0], fmaf(rv[1], rv[1], fmaf(rv[2], rv[2], fmaf(rv[3], rv[3], rr))));
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      const f32x4 xv = ok ? *reinterpret_cast<const f32x4*>(xrow[qb] + col) : zero4;
      xx[qb] = fmaf(xv[0], xv[0], fmaf(xv[1], xv[1], fmaf(xv[2], xv[2], fmaf(xv[3], xv[3], xx[qb]))));
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(rv[e], xv[e], acc[qb], 0, 0, 0);
    }
  }
  // accumulator: register e <-> reference m0 + 4 g + e, lane column <-> query n0 + 16 qb + li
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const int n = n0 + qb * 16 + li;
    if (n < np) *reinterpret_cast<f32x4*>(gpart + ((int64_t)sp * np + n) * mp + m0 + 4 * g) = acc[qb];
  }
  rr += __shfl_xor(rr, 16, 64); rr += __shfl_xor(rr, 32, 64);       // the four column groups of a row
  if (g == 0 && ng == 0) rrpart[(int64_t)sp * mp + m0 + li] = rr;
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    float v = xx[qb];
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    const int n = n0 + qb * 16 + li;
    if (g == 0 && mq == 0 && wid == 0 && n < np) xxpart[(int64_t)sp * np + n] = v;
  }
}


// ---- squared distances by direct differences (SPARSE only) -------------------------------------------------------
// The radius test of the sparse variant compares distances that can be SMALL next to |x|, |r| (a noisy copy of a
// reference at t = 1 in the calibration; a query inside a reference's ball): there the Gram form's cancellation costs
// 1e-4 relative, so SPARSE keeps sum_j (x_j - r_j)^2.  Not on the headline path (kernel_fast is RBF).
constexpr int kQ = 8, kR = 8;
__global__ void __launch_bounds__(kThreads)
k_dist2_tile(const float* __restrict__ xq, const float* __restrict__ R, int N, int M, int64_t D, float* __restrict__ d2) {
  __shared__ float red[4][kR * kQ];
  const int m0 = blockIdx.x * kR, n0 = blockIdx.y * kQ;
  const int64_t d4 = D / 4;
  float acc[kR][kQ];
#pragma unroll
  for (int i = 0; i < kR; ++i)
#pragma unroll
    for (int q = 0; q < kQ; ++q) acc[i][q] = 0.f;
  for (int64_t j = threadIdx.x; j < d4; j += kThreads) {
    float4 r[kR], a[kQ];
#pragma unroll
    for (int i = 0; i < kR; ++i) r[i] = reinterpret_cast<const float4*>(R + (int64_t)min(m0 + i, M - 1) * D)[j];
#pragma unroll
    for (int q = 0; q < kQ; ++q) a[q] = reinterpret_cast<const float4*>(xq + (int64_t)min(n0 + q, N - 1) * D)[j];
#pragma unroll
    for (int i = 0; i < kR; ++i)
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        const float e0 = a[q].x - r[i].x, e1 = a[q].y - r[i].y, e2 = a[q].z - r[i].z, e3 = a[q].w - r[i].w;
        acc[i][q] = fmaf(e0, e0, fmaf(e1, e1, fmaf(e2, e2, fmaf(e3, e3, acc[i][q]))));
      }
  }
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < kR; ++i)
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      const float sm = wave_sum(acc[i][q]);
      if (lane == 0) red[wid][i * kQ + q] = sm;
    }
  __syncthreads();
  if (threadIdx.x < kR * kQ) {
    const int i = threadIdx.x / kQ, q = threadIdx.x % kQ;
    if (m0 + i < M && n0 + q < N)
      d2[(int64_t)(n0 + q) * M + m0 + i] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// distance of (query n, reference m) from the slice partials, slices summed in index order
__device__ __forceinline__ float dist_from_partials(const float* __restrict__ gpart, const float* __restrict__ rrpart, float xx,
                                                    int splits, int np, int mp, int n, int m, const float* __restrict__ d2_direct = nullptr,
                                                    int M = 0) {
  if (d2_direct) return sqrtf(d2_direct[(int64_t)n * M + m]);      // SPARSE: direct-difference distances
  float G = 0.f, rr = 0.f;
  for (int s = 0; s < splits; ++s) { G += gpart[((int64_t)s * np + n) * mp + m]; rr += rrpart[(int64_t)s * mp + m]; }
  float d2 = (xx + rr) - 2.f * G;
  d2 = d2 < 0.f ? 0.f : d2;                                        // rounding below zero; a NaN stays a NaN
  return sqrtf(d2);
}
__device__ __forceinline__ float weight_of(float dist, int weight_fn, float inv_two_sigma_sq, float radius) {
  if (weight_fn == SDN_REPEL_RBF) return expf(-dist * inv_two_sigma_sq);
  return (dist < radius) ? fmaxf(radius / dist - 1.f, 0.f) : 0.f;   // dist NaN -> not a neighbour
}

// ---- weights, denominator, gate: one workgroup per query --------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
k_weights(const float* __restrict__ gpart, const float* __restrict__ rrpart, const float* __restrict__ xxpart, int splits, int np,
          int mp, int M, int weight_fn, float inv_two_sigma_sq, float radius, float eps, float gate, float* __restrict__ w,
          float* __restrict__ den_ws, float* __restrict__ out_den, int32_t* __restrict__ out_isneg, float* __restrict__ d_out,
          const float* __restrict__ d2) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  float xx = 0.f;
  if (!d2) for (int s = 0; s < splits; ++s) xx += xxpart[(int64_t)s * np + n];
  float sum = 0.f;
  for (int m = threadIdx.x; m < mp; m += kThreads) {
    float wv = 0.f;
    if (m < M) {
      const float dist = dist_from_partials(gpart, rrpart, xx, splits, np, mp, n, m, d2, M);
      if (d_out) d_out[(int64_t)n * M + m] = dist;
      wv = weight_of(dist, weight_fn, inv_two_sigma_sq, radius);
    }
    if (w) w[(int64_t)n * mp + m] = wv;
    sum += wv;
  }
  const float tot = block_sum<4>(sum, red);
  if (threadIdx.x == 0) {
    const float den = (weight_fn == SDN_REPEL_RBF) ? tot + eps : tot;
    if (den_ws) den_ws[n] = den;
    if (out_den) out_den[n] = den;
    if (out_isneg) out_isneg[n] = (weight_fn == SDN_REPEL_RBF) ? (den > gate ? 1 : 0) : (tot != 0.f ? 1 : 0);
  }
}

// ---- weighted sum of reference rows + the update of x ------------------------------------------------------------
// Workgroup = (64 columns, 64-query group).  MFMA orientation: A[i = query][k = ref], B[k = ref][j]; a lane (j = lane & 15,
// g = lane >> 4) loads 4 consecutive columns of reference row m + g and the four elements feed four accumulators
// (accumulator e holds column c0 + 4 j + e).  Wave w takes the reference quads w, w + 4, w + 8, ...
struct WsumArgs {
  const float* w; const float* den; const float* R; float* x; const float* xq; float* out_neg;
  int N, M; int64_t D; int np, mp, weight_fn; float scale;
  // self-service weights (N <= 4): slice partials of the Gram sweep + the parameters of k_weights
  const float* gpart; const float* rrpart; const float* xxpart; int splits;
  float inv_two_sigma_sq, radius, eps, gate; float* out_den; int32_t* out_isneg;
};

template <int NQB, bool SELF>
__global__ void __launch_bounds__(kThreads)
k_wsum(const WsumArgs a) {
  extern __shared__ float lds[];                                   // [4 waves][16 NQB][64] partial tiles (+ SELF: weights)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lj = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * 64;
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  const float* wsrc = a.w;
  int wld = a.mp;
  float* wself = lds + 4 * 16 * NQB * 64;                          // SELF: w[n][mp] for n < N, and den[n] behind it
  if constexpr (SELF) {
    __shared__ float red[4];
    for (int n = 0; n < a.N; ++n) {
      float xx = 0.f;
      for (int s = 0; s < a.splits; ++s) xx += a.xxpart[(int64_t)s * a.np + n];
      float sum = 0.f;
      for (int m = threadIdx.x; m < a.mp; m += kThreads) {
        float wv = 0.f;
        if (m < a.M)
          wv = weight_of(dist_from_partials(a.gpart, a.rrpart, xx, a.splits, a.np, a.mp, n, m), a.weight_fn, a.inv_two_sigma_sq, a.radius);
        wself[n * a.mp + m] = wv;
        sum += wv;
      }
      const float tot = block_sum<4>(sum, red);
      if (threadIdx.x == 0) {
        const float den = (a.weight_fn == SDN_REPEL_RBF) ? tot + a.eps : tot;
        wself[a.N * a.mp + n] = den;
        if (blockIdx.x == 0) {
          if (a.out_den) a.out_den[n] = den;
          if (a.out_isneg) a.out_isneg[n] = (a.weight_fn == SDN_REPEL_RBF) ? (den > a.gate ? 1 : 0) : (tot != 0.f ? 1 : 0);
        }
      }
    }
    __syncthreads();
  }
  f32x4 acc[NQB][4];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[qb][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool col_ok = c0 + 4 * lj < a.D;
  const float* rcol = a.R + c0 + 4 * lj;
  for (int m = 4 * wid; m < a.M; m += 16) {
    const int mr = m + g;
    const f32x4 rv = (col_ok && mr < a.M) ? *reinterpret_cast<const f32x4*>(rcol + (int64_t)mr * a.D) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      const int n = n0 + qb * 16 + lj;                               // A[i = query lj][k = g] = w[n][m + g]  (padded rows are zero)
      float wv;
      if constexpr (SELF) wv = (n < a.N && mr < a.mp) ? wself[n * a.mp + mr] : 0.f;
      else wv = (n < a.N && mr < a.mp) ? wsrc[(int64_t)n * wld + mr] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[qb][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, rv[e], acc[qb][e], 0, 0, 0);
    }
  }
  // accumulator (qb, e): register r <-> query n0 + 16 qb + 4 g + r, lane column <-> column c0 + 4 lj + e
  float* tile = lds + (int64_t)wid * 16 * NQB * 64;
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<f32x4*>(tile + (qb * 16 + 4 * g + r) * 64 + 4 * lj) =
          (f32x4){acc[qb][0][r], acc[qb][1][r], acc[qb][2][r], acc[qb][3][r]};
  __syncthreads();
  for (int it = threadIdx.x; it < 16 * NQB * 16; it += kThreads) {
    const int q = it >> 4, c4 = (it & 15) * 4;
    const int n = n0 + q;
    if (n >= a.N || c0 + c4 >= a.D) continue;
    f32x4 s = *reinterpret_cast<const f32x4*>(lds + (0 * 16 * NQB + q) * 64 + c4);
#pragma unroll
    for (int w_ = 1; w_ < 4; ++w_) s += *reinterpret_cast<const f32x4*>(lds + ((int64_t)w_ * 16 * NQB + q) * 64 + c4);
    float den;
    if constexpr (SELF) den = wself[a.N * a.mp + n]; else den = a.den[n];
    f32x4* xp = reinterpret_cast<f32x4*>(a.x + (int64_t)n * a.D + c0 + c4);
    f32x4 xv = *xp, gg;
    if (a.weight_fn == SDN_REPEL_RBF) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { gg[e] = s[e] / den; xv[e] -= a.scale * gg[e]; }
    } else {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(a.xq + (int64_t)n * a.D + c0 + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gg[e] = qv[e] * den - s[e]; xv[e] += a.scale * gg[e]; }
    }
    *xp = xv;
    if (a.out_neg) *reinterpret_cast<f32x4*>(a.out_neg + (int64_t)n * a.D + c0 + c4) = gg;
  }
}

// ---- calibration tail: beta[n] = sum_m exp(-dist / 2 sigma^2) + eps comes out of k_weights (den) -----------------

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int check_params(const sdn_repel_params* p) {
  if (!p) return SDN_E_INVALID;
  if (p->n_query < 0 || p->n_ref < 0 || p->channels <= 0 || p->hw <= 0) return SDN_E_INVALID;
  const int64_t D = (int64_t)p->channels * p->hw;
  if (D & 3) return SDN_E_INVALID;
  if (p->weight_fn != SDN_REPEL_RBF && p->weight_fn != SDN_REPEL_SPARSE) return SDN_E_INVALID;
  if (p->qnorm != SDN_QNORM_NONE && p->qnorm != SDN_QNORM_CHANNEL) return SDN_E_INVALID;
  if (p->weight_fn == SDN_REPEL_RBF && !(p->sigma > 0.f)) return SDN_E_INVALID;
  return SDN_OK;
}

// Shared front half: (qnorm) + the Gram sweep.  Returns the pointer the distances were computed from.
inline const float* run_gram(const sdn_repel_params* p, const Plan& pl, const float* x, const float* R, char* ws,
                             hipStream_t st) {
  const int N = p->n_query, M = p->n_ref;
  const int64_t D = (int64_t)p->channels * p->hw;
  const float* xq = x;
  if (p->qnorm == SDN_QNORM_CHANNEL) {
    float* q = reinterpret_cast<float*>(ws + pl.off_xq);
    hipLaunchKernelGGL(k_qnorm, dim3((p->hw + kThreads - 1) / kThreads, N), dim3(kThreads), 0, st, x, q,
                       p->channels, p->hw);
    xq = q;
  }
  if (M > 0 && p->weight_fn == SDN_REPEL_SPARSE) {
    hipLaunchKernelGGL(k_dist2_tile, dim3((M + kR - 1) / kR, (N + kQ - 1) / kQ), dim3(kThreads), 0, st, xq, R, N, M, D,
                       reinterpret_cast<float*>(ws + pl.off_d2));
  } else if (M > 0) {
    float* gp = reinterpret_cast<float*>(ws + pl.off_g);
    float* rr = reinterpret_cast<float*>(ws + pl.off_rr);
    float* xx = reinterpret_cast<float*>(ws + pl.off_xx);
    const dim3 grid(pl.mquads, pl.splits, pl.ngroups);
    if (N <= 16) hipLaunchKernelGGL((k_gram<1>), grid, dim3(kThreads), 0, st, xq, R, N, M, D, pl.cps, pl.np, pl.mp, gp, rr, xx);
    else if (N <= 32) hipLaunchKernelGGL((k_gram<2>), grid, dim3(kThreads), 0, st, xq, R, N, M, D, pl.cps, pl.np, pl.mp, gp, rr, xx);
    else hipLaunchKernelGGL((k_gram<4>), grid, dim3(kThreads), 0, st, xq, R, N, M, D, pl.cps, pl.np, pl.mp, gp, rr, xx);
  }
  return xq;
}

}  // namespace

extern "C" {

size_t sdn_repel_workspace_bytes(int32_t n_query, int32_t n_ref, int32_t channels, int32_t hw) {
  if (n_query < 0 || n_ref < 0 || channels <= 0 || hw <= 0) return 0;
  return make_plan(n_query, n_ref, (int64_t)channels * hw).total;
}

int sdn_repel_apply(const sdn_repel_params* p, float* x, const float* R, float* out_neg, float* out_den,
                    int32_t* out_isneg, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_params(p);
  if (rc != SDN_OK) return rc;
  const int N = p->n_query, M = p->n_ref;
  if (N == 0) return SDN_OK;
  if (!x || !workspace || (M > 0 && !R) || !aligned16(x) || !aligned16(R) || (out_neg && !aligned16(out_neg)) ||
      !aligned16(workspace))
    return SDN_E_INVALID;
  const int64_t D = (int64_t)p->channels * p->hw;
  const Plan pl = make_plan(N, M, D);
  if (workspace_bytes < pl.total) return SDN_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* ws = static_cast<char*>(workspace);
  float* w = reinterpret_cast<float*>(ws + pl.off_w);
  float* den = reinterpret_cast<float*>(ws + pl.off_den);
  const float* gp = reinterpret_cast<const float*>(ws + pl.off_g);
  const float* rr = reinterpret_cast<const float*>(ws + pl.off_rr);
  const float* xx = reinterpret_cast<const float*>(ws + pl.off_xx);
  const float* xq = run_gram(p, pl, x, R, ws, st);
  const int splits = M > 0 ? pl.splits : 0;                        // empty reference set: den = eps (RBF) / 0 (SPARSE), neg = 0
  const float i2s = 1.f / (2.f * p->sigma * p->sigma);
  // one prompt (the reference's shape) .. four: every k_wsum workgroup derives the weights itself -> no k_weights launch
  const float* d2 = (p->weight_fn == SDN_REPEL_SPARSE && M > 0) ? reinterpret_cast<const float*>(ws + pl.off_d2) : nullptr;
  const bool self = N <= 4 && !d2 && (size_t)(4 * 16 * 64 + N * pl.mp + N) * 4 <= 64 * 1024;
  if (!self)
    hipLaunchKernelGGL(k_weights, dim3(N), dim3(kThreads), 0, st, gp, rr, xx, splits, pl.np, pl.mp, M, p->weight_fn, i2s,
                       p->radius, p->epsilon, p->gate, w, den, out_den, out_isneg, (float*)nullptr, d2);
  WsumArgs a{w, den, R, x, xq, out_neg, N, M, D, pl.np, pl.mp, p->weight_fn, p->scale,
             gp, rr, xx, splits, i2s, p->radius, p->epsilon, p->gate, out_den, out_isneg};
  const dim3 grid((unsigned)((D + 63) / 64), (unsigned)pl.ngroups);
  if (self) {
    const size_t lds = (size_t)(4 * 16 * 64 + N * pl.mp + N) * 4;
    hipLaunchKernelGGL((k_wsum<1, true>), grid, dim3(kThreads), lds, st, a);
  } else if (N <= 16) {
    hipLaunchKernelGGL((k_wsum<1, false>), grid, dim3(kThreads), (size_t)4 * 16 * 1 * 64 * 4, st, a);
  } else if (N <= 32) {
    hipLaunchKernelGGL((k_wsum<2, false>), grid, dim3(kThreads), (size_t)4 * 16 * 2 * 64 * 4, st, a);
  } else {
    hipLaunchKernelGGL((k_wsum<4, false>), grid, dim3(kThreads), (size_t)4 * 16 * 4 * 64 * 4, st, a);
  }
  return sdn_launch_status();
}

int sdn_repel_calibrate(const sdn_repel_params* p, const float* queries, const float* R, float* out,
                        void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_params(p);
  if (rc != SDN_OK) return rc;
  const int N = p->n_query, M = p->n_ref;
  if (N == 0) return SDN_OK;
  if (!queries || !R || !out || !workspace || M <= 0 || !aligned16(queries) || !aligned16(R) ||
      !aligned16(workspace))
    return SDN_E_INVALID;
  const int64_t D = (int64_t)p->channels * p->hw;
  const Plan pl = make_plan(N, M, D);
  if (workspace_bytes < pl.total) return SDN_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* ws = static_cast<char*>(workspace);
  run_gram(p, pl, queries, R, ws, st);
  const float* gp = reinterpret_cast<const float*>(ws + pl.off_g);
  const float* rr = reinterpret_cast<const float*>(ws + pl.off_rr);
  const float* xx = reinterpret_cast<const float*>(ws + pl.off_xx);
  // RBF: beta[n] = the denominator (sum of the weights + eps); SPARSE: the pairwise distances themselves
  if (p->weight_fn == SDN_REPEL_RBF)
    hipLaunchKernelGGL(k_weights, dim3(N), dim3(kThreads), 0, st, gp, rr, xx, pl.splits, pl.np, pl.mp, M, SDN_REPEL_RBF,
                       1.f / (2.f * p->sigma * p->sigma), 0.f, p->epsilon, 0.f, (float*)nullptr, (float*)nullptr, out,
                       (int32_t*)nullptr, (float*)nullptr, (const float*)nullptr);
  else
    hipLaunchKernelGGL(k_weights, dim3(N), dim3(kThreads), 0, st, gp, rr, xx, pl.splits, pl.np, pl.mp, M, SDN_REPEL_SPARSE,
                       1.f, 0.f, 0.f, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr, (int32_t*)nullptr, out,
                       reinterpret_cast<const float*>(ws + pl.off_d2));
  return sdn_launch_status();
}

}  // extern "C"
