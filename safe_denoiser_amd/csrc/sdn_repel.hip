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
//   k_weights d2 = |x|^2 + |r|^2 - 2 G (slices summed in fixed order), w[n,m] (RBF or SPARSE), per-block weight sums
//   k_wsum    neg[n,j] = sum_m w[n,m] R[m,j] for a (<=64 queries x 64 columns) block, the eight waves taking interleaved
//             reference rows and combining through LDS in fixed order; epilogue fused: den[n] and is_negation[n], neg / den,
//             the in-place update of x, the optional negative-score output.
// The Gram form's cancellation is harmless here: a query is never close to a reference in units of |x|^2 + |r|^2 (the
// reference's own torch.cdist takes the same form above 25 rows); fp32 error in d is ~1e-7 * (|x|^2 + |r|^2) / (2 d).
// Results are deterministic (no float atomics; every sum has a fixed order).
#include "sdn_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxSplits = 128;

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct Plan {
  int np;                 // N rounded up to 16 (rows of the padded per-query buffers)
  int mp;                 // M rounded up to 64
  int mquads;             // 64-reference blocks
  int splits, cps;        // column slices of the Gram sweep, columns per slice (multiple of 16)
  int ngroups;            // 64-query groups
  size_t off_xq, off_g, off_rr, off_xx, off_w, off_den, off_d2, total;   // off_den: per-block weight sums [np][mquads]
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

inline Plan make_plan(int N, int M, int64_t D) {
  Plan p;
  p.np = N <= 16 ? 16 : (N <= 32 ? 32 : (N + 63) / 64 * 64);          // whole 16 x NQB query tiles (k_wsum's vector weight loads)
  p.mp = (M + 63) / 64 * 64; if (p.mp < 64) p.mp = 64;
  p.mquads = p.mp / 64;
  p.ngroups = (N + 63) / 64; if (p.ngroups < 1) p.ngroups = 1;
  // column slices: enough of them for ~2 (small N: ~4) workgroups per CU, at least 64 columns each
  const int64_t target = N <= 16 ? 1024 : 512;
  int64_t want = (target + (int64_t)p.mquads * p.ngroups - 1) / ((int64_t)p.mquads * p.ngroups);
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
  p.off_den = o; o += align256((size_t)p.np * p.mquads * 4);
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
// Workgroup = (64-reference block, column slice, 64-query group), 4 waves.  Per 64-column step the workgroup stages the
// [64 refs x 64 cols] tile of R and the [16 NQB queries x 64 cols] tile of x through LDS -- global loads are whole
// 256-byte row segments, 4 rows per wave-instruction (the fragment shape an MFMA wants, 16 rows x 64 B, wastes DRAM
// bursts on a 34 MB streaming read) -- and wave w contracts reference rows 16 w .. 16 w + 15 against all queries.  The
// next step's tiles are loaded into registers before the current step's MFMAs (two workgroups per CU keep ~64 KB in flight).
// MFMA orientation: A[i = ref][k], B[k][j = query]; lane (i | j = lane & 15, g = lane >> 4) reads 4 consecutive columns
// (ds_read_b128) and the four elements feed four consecutive MFMAs: MFMA e of sub-step u sees column 16 u + 4 g + e at
// k = g (any assignment of columns to k slots is a valid contraction as long as A and B agree).
constexpr int kGramLd = 68;                                        // LDS row stride in floats (64 + 4: rows 4 banks apart)
template <int NQB>
__global__ void __launch_bounds__(kThreads)
k_gram(const float* __restrict__ xq, const float* __restrict__ R, int N, int M, int64_t D, int cps, int np, int mp,
       float* __restrict__ gpart, float* __restrict__ rrpart, float* __restrict__ xxpart) {
  __shared__ __attribute__((aligned(16))) float sr[2][64 * kGramLd], sx[2][16 * NQB * kGramLd];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int mq = blockIdx.x, sp = blockIdx.y, ng = blockIdx.z;
  const int m0 = mq * 64, n0 = ng * 64;
  const int64_t c_lo = (int64_t)sp * cps;
  const int64_t c_hi = c_lo + cps < D ? c_lo + cps : D;
  // staging role: thread -> (row = tid / 16 + 16 k, 16-byte chunk tid % 16) for k = 0..3 (R) and k < NQB (x)
  const int srow = tid >> 4, sc4 = (tid & 15) * 4;
  const float* rsrc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int m = m0 + srow + 16 * k;
    rsrc[k] = R + (int64_t)(m < M ? m : M - 1) * D + sc4;
  }
  const float* xsrc[NQB];
#pragma unroll
  for (int k = 0; k < NQB; ++k) {
    const int n = n0 + srow + 16 * k;
    xsrc[k] = xq + (int64_t)(n < N ? n : N - 1) * D + sc4;
  }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) acc[qb] = zero4;
  float rr = 0.f, xx[NQB];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) xx[qb] = 0.f;
  // Two-stage LDS ring, one barrier per 64-column step; the next step's tiles are loaded into registers before this step's
  // MFMAs.  (Issuing the loads of the whole slice up front was measured and is SLOWER, 26 vs 21 us at 64 queries: the
  // sweep is not short of bytes in flight -- a single-prompt call already runs at the ~10 us a 34 MB cold read costs.)
  f32x4 lr[4], lx[NQB];
  auto g_load = [&](int64_t c) {
    const bool ok = c + sc4 < c_hi;                                 // D % 4 == 0: a float4 is inside or outside as a whole
#pragma unroll
    for (int k = 0; k < 4; ++k) lr[k] = ok ? *reinterpret_cast<const f32x4*>(rsrc[k] + c) : zero4;
#pragma unroll
    for (int k = 0; k < NQB; ++k) lx[k] = ok ? *reinterpret_cast<const f32x4*>(xsrc[k] + c) : zero4;
  };
  auto s_store = [&](int buf) {
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(&sr[buf][(srow + 16 * k) * kGramLd + sc4]) = lr[k];
#pragma unroll
    for (int k = 0; k < NQB; ++k) *reinterpret_cast<f32x4*>(&sx[buf][(srow + 16 * k) * kGramLd + sc4]) = lx[k];
  };
  g_load(c_lo);
  s_store(0);
  __syncthreads();
  int buf = 0;
  for (int64_t c = c_lo; c < c_hi; c += 64, buf ^= 1) {
    const bool more = c + 64 < c_hi;
    if (more) g_load(c + 64);                                       // in flight during this step's MFMAs
    const float* ar = &sr[buf][(wid * 16 + li) * kGramLd + 4 * g];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(ar + 16 * u);
      rr = fmaf(av[0], av[0], fmaf(av[1], av[1], fmaf(av[2], av[2], fmaf(av[3], av[3], rr))));
      f32x4 bv[NQB];
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        bv[qb] = *reinterpret_cast<const f32x4*>(&sx[buf][(qb * 16 + li) * kGramLd + 16 * u + 4 * g]);
        xx[qb] = fmaf(bv[qb][0], bv[qb][0], fmaf(bv[qb][1], bv[qb][1], fmaf(bv[qb][2], bv[qb][2], fmaf(bv[qb][3], bv[qb][3], xx[qb]))));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)                                   // e outer: consecutive MFMAs go to different accumulators
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) acc[qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[qb][e], acc[qb], 0, 0, 0);
    }
    if (more) s_store(buf ^ 1);                                     // the other stage: last read before the previous barrier
    __syncthreads();
  }
  // accumulator: register e <-> reference m0 + 16 w + 4 g + e, lane column <-> query n0 + 16 qb + li
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const int n = n0 + qb * 16 + li;
    if (n < np) *reinterpret_cast<f32x4*>(gpart + ((int64_t)sp * np + n) * mp + m0 + wid * 16 + 4 * g) = acc[qb];
  }
  rr += __shfl_xor(rr, 16, 64); rr += __shfl_xor(rr, 32, 64);       // the four column groups of a row
  if (g == 0 && ng == 0) rrpart[(int64_t)sp * mp + m0 + wid * 16 + li] = rr;
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

__device__ __forceinline__ float weight_of(float dist, int weight_fn, float inv_two_sigma_sq, float radius) {
  if (weight_fn == SDN_REPEL_RBF) return expf(-dist * inv_two_sigma_sq);
  return (dist < radius) ? fmaxf(radius / dist - 1.f, 0.f) : 0.f;   // dist NaN -> not a neighbour
}

// ---- weights: one workgroup per (query, 64-reference block); 4 threads share a reference and split the slices ------
__global__ void __launch_bounds__(kThreads)
k_weights(const float* __restrict__ gpart, const float* __restrict__ rrpart, const float* __restrict__ xxpart, int splits, int np,
          int mp, int M, int weight_fn, float inv_two_sigma_sq, float radius, float* __restrict__ w, float* __restrict__ denpart,
          float* __restrict__ d_out, const float* __restrict__ d2) {
  __shared__ float pg[4][64], pr[4][64], red[4];
  const int n = blockIdx.x, mq = blockIdx.y;
  const int ml = threadIdx.x & 63, sg = threadIdx.x >> 6;           // sg = wave = slice group (slices sg, sg + 4, ...)
  const int m = mq * 64 + ml;
  float G = 0.f, rr = 0.f, xx = 0.f;
  if (!d2) {
#pragma unroll 4
    for (int s = sg; s < splits; s += 4) { G += gpart[((int64_t)s * np + n) * mp + m]; rr += rrpart[(int64_t)s * mp + m]; }
    float xs = 0.f;
    for (int s = threadIdx.x; s < splits; s += kThreads) xs += xxpart[(int64_t)s * np + n];
    xx = block_sum<4>(xs, red);                                    // fixed butterfly order: deterministic
  }
  pg[sg][ml] = G; pr[sg][ml] = rr;
  __syncthreads();
  float wv = 0.f;
  if (sg == 0) {
    if (m < M) {
      float dist;
      if (d2) {
        dist = sqrtf(d2[(int64_t)n * M + m]);                       // SPARSE: direct-difference distances
      } else {
        const float Gs = ((pg[0][ml] + pg[1][ml]) + pg[2][ml]) + pg[3][ml], rs = ((pr[0][ml] + pr[1][ml]) + pr[2][ml]) + pr[3][ml];
        float dd = (xx + rs) - 2.f * Gs;
        dd = dd < 0.f ? 0.f : dd;                                   // rounding below zero; a NaN stays a NaN
        dist = sqrtf(dd);
      }
      if (d_out) d_out[(int64_t)n * M + m] = dist;
      wv = weight_of(dist, weight_fn, inv_two_sigma_sq, radius);
    }
    if (w) w[(int64_t)m * np + n] = wv;                             // transposed [mp][np]: k_wsum reads 16 consecutive queries
  }
  const float tot = block_sum<4>(wv, red);
  if (threadIdx.x == 0 && denpart) denpart[(int64_t)n * gridDim.y + mq] = tot;
}

// den[n] = sum of the per-block weight sums (+ eps for RBF), gate; one thread per query
__global__ void k_den(const float* __restrict__ denpart, int N, int mquads, int weight_fn, float eps, float gate,
                      float* __restrict__ den_ws, float* __restrict__ out_den, int32_t* __restrict__ out_isneg) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float tot = 0.f;
  for (int q = 0; q < mquads; ++q) tot += denpart[(int64_t)n * mquads + q];
  const float den = (weight_fn == SDN_REPEL_RBF) ? tot + eps : tot;
  if (den_ws) den_ws[n] = den;
  if (out_den) out_den[n] = den;
  if (out_isneg) out_isneg[n] = (weight_fn == SDN_REPEL_RBF) ? (den > gate ? 1 : 0) : (tot != 0.f ? 1 : 0);
}

// ---- weighted sum of reference rows + the update of x ------------------------------------------------------------
// Workgroup = (64 columns, 64-query group), 8 waves.  MFMA orientation: A[i = query][k = ref], B[k = ref][j]; a lane
// (j = lane & 15, g = lane >> 4) loads 4 consecutive columns of reference row m + g and the four elements feed four
// accumulators (accumulator e holds column c0 + 4 j + e).  Wave w takes the reference quads w, w + 8, w + 16, ...; the
// eight partial tiles are summed through LDS in a fixed order ((w, w + 4) pairs, then 0..3).  den[n] is finished here from
// the per-block weight sums, so a projection is three launches: k_gram, k_weights, k_wsum.
constexpr int kWsumThreads = 512;
struct WsumArgs {
  const float* w; const float* denpart; const float* R; float* x; const float* xq; float* out_neg;
  int N, M; int64_t D; int np, mp, mquads, weight_fn; float scale, eps, gate; float* out_den; int32_t* out_isneg;
};

template <int NQB>
__global__ void __launch_bounds__(kWsumThreads)
k_wsum(const WsumArgs a) {
  extern __shared__ float lds[];                                   // [4][16 NQB][64] partial tiles
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lj = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * 64;
  const int64_t c0 = (int64_t)blockIdx.x * 64;
  // Query mapping: tile qb, MFMA row i  <->  query n0 + NQB * i + qb, so that the NQB weights a lane needs for one reference
  // (A[i = lj][k = g] of the NQB tiles) are NQB consecutive floats of the transposed weight matrix: one load.
  typedef float wvec __attribute__((ext_vector_type(NQB)));
  f32x4 acc[NQB][4];
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[qb][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool col_ok = c0 + 4 * lj < a.D;
  const float* rcol = a.R + c0 + 4 * lj;
  const int nq0 = n0 + NQB * lj;                                   // first of this lane's NQB queries (np covers whole tiles)
  // U reference quads per wave per pass, ALL their loads issued before the first MFMA (the sweep is 34 MB: it has to be in
  // flight at once); 8 waves x 17 quads x 4 = 544 references per pass (64-query tiles: 9 quads = 288, registers).
  constexpr int U = NQB == 4 ? 9 : 17;
  for (int mb = 0; mb < a.M; mb += 32 * U) {
    f32x4 rv[U];
    wvec wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int mr = mb + 32 * u + 4 * wid + g;
      const bool ok = mr < a.M;
      rv[u] = (col_ok && ok) ? *reinterpret_cast<const f32x4*>(rcol + (int64_t)mr * a.D) : (f32x4){0.f, 0.f, 0.f, 0.f};
      wv[u] = ok ? *reinterpret_cast<const wvec*>(a.w + (int64_t)mr * a.np + nq0) : (wvec)(0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (mb + 32 * u + 4 * wid >= a.M) break;                      // wave-uniform
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
          const float wq = wv[u][qb];
          acc[qb][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq, rv[u][e], acc[qb][e], 0, 0, 0);
        }
    }
  }
  // accumulator (qb, e): register r <-> query n0 + NQB (4 g + r) + qb, lane column <-> column c0 + 4 lj + e
  float* tile = lds + (int64_t)(wid & 3) * 16 * NQB * 64;
  auto put = [&]() {
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<f32x4*>(tile + (NQB * (4 * g + r) + qb) * 64 + 4 * lj) =
            (f32x4){acc[qb][0][r], acc[qb][1][r], acc[qb][2][r], acc[qb][3][r]};
  };
  if (wid >= 4) put();
  __syncthreads();
  if (wid < 4) {
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(tile + (NQB * (4 * g + r) + qb) * 64 + 4 * lj);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[qb][e][r] += o[e];
      }
  }
  __syncthreads();
  if (wid < 4) put();
  __syncthreads();
  for (int it = threadIdx.x; it < 16 * NQB * 16; it += kWsumThreads) {
    const int q = it >> 4, c4 = (it & 15) * 4;
    const int n = n0 + q;
    if (n >= a.N || c0 + c4 >= a.D) continue;
    f32x4 s = *reinterpret_cast<const f32x4*>(lds + (0 * 16 * NQB + q) * 64 + c4);
#pragma unroll
    for (int w_ = 1; w_ < 4; ++w_) s += *reinterpret_cast<const f32x4*>(lds + ((int64_t)w_ * 16 * NQB + q) * 64 + c4);
    float tot = 0.f;
    for (int mq = 0; mq < a.mquads; ++mq) tot += a.denpart[(int64_t)n * a.mquads + mq];
    const float den = (a.weight_fn == SDN_REPEL_RBF) ? tot + a.eps : tot;
    if (blockIdx.x == 0 && c4 == 0) {
      if (a.out_den) a.out_den[n] = den;
      if (a.out_isneg) a.out_isneg[n] = (a.weight_fn == SDN_REPEL_RBF) ? (den > a.gate ? 1 : 0) : (tot != 0.f ? 1 : 0);
    }
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
  const float* d2 = (p->weight_fn == SDN_REPEL_SPARSE && M > 0) ? reinterpret_cast<const float*>(ws + pl.off_d2) : nullptr;
  hipLaunchKernelGGL(k_weights, dim3(N, pl.mquads), dim3(kThreads), 0, st, gp, rr, xx, splits, pl.np, pl.mp, M, p->weight_fn, i2s,
                     p->radius, w, den, (float*)nullptr, d2);
  WsumArgs a{w, den, R, x, xq, out_neg, N, M, D, pl.np, pl.mp, pl.mquads, p->weight_fn, p->scale, p->epsilon, p->gate,
             out_den, out_isneg};
  const dim3 grid((unsigned)((D + 63) / 64), (unsigned)pl.ngroups);
  if (N <= 16) hipLaunchKernelGGL((k_wsum<1>), grid, dim3(kWsumThreads), (size_t)4 * 16 * 1 * 64 * 4, st, a);
  else if (N <= 32) hipLaunchKernelGGL((k_wsum<2>), grid, dim3(kWsumThreads), (size_t)4 * 16 * 2 * 64 * 4, st, a);
  else hipLaunchKernelGGL((k_wsum<4>), grid, dim3(kWsumThreads), (size_t)4 * 16 * 4 * 64 * 4, st, a);
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
  float* den = reinterpret_cast<float*>(ws + pl.off_den);
  if (p->weight_fn == SDN_REPEL_RBF) {
    hipLaunchKernelGGL(k_weights, dim3(N, pl.mquads), dim3(kThreads), 0, st, gp, rr, xx, pl.splits, pl.np, pl.mp, M, SDN_REPEL_RBF,
                       1.f / (2.f * p->sigma * p->sigma), 0.f, (float*)nullptr, den, (float*)nullptr, (const float*)nullptr);
    hipLaunchKernelGGL(k_den, dim3((N + 255) / 256), dim3(256), 0, st, den, N, pl.mquads, SDN_REPEL_RBF, p->epsilon, 0.f,
                       (float*)nullptr, out, (int32_t*)nullptr);
  } else {
    hipLaunchKernelGGL(k_weights, dim3(N, pl.mquads), dim3(kThreads), 0, st, gp, rr, xx, pl.splits, pl.np, pl.mp, M, SDN_REPEL_SPARSE,
                       1.f, 0.f, (float*)nullptr, (float*)nullptr, out, reinterpret_cast<const float*>(ws + pl.off_d2));
  }
  return sdn_launch_status();
}

}  // extern "C"
