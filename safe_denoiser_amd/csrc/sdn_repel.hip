// Repellency projection (rows R1-R5): kernel-weighted mean of proj_ref rows + in-place axpy, batched over
// N queries, fp32, HBM-bound.
//
// Data layout: proj_ref R is [M, D] fp32 row-major (D = C*H*W, the reference's NCHW flattening), 64 KiB per
// row for SD-v1.4.  Algorithmic traffic per call = ONE read of R (M*D*4 B) + x in/out.  The projection needs
// R twice (distances, then the weighted sum); M*D*4 = 33.75 MB (135 MB SD-v3) stays resident in the 256 MiB
// Infinity Cache between the two passes, so the second pass does not return to HBM.
//
//   k_qnorm      (SD-v3 only) xq = x / ||x||_channel per pixel
//   k_dist2      d2[n,m] = sum_j (xq[n,j] - R[m,j])^2            one workgroup per (ref row, 8-query chunk):
//                coalesced float4 sweep of the row, wave-shuffle + LDS reduction
//   k_weights    w[n,m] from d2 (RBF or SPARSE), den[n], is_negation[n]   (tiny; one workgroup per query)
//   k_wsum       part[s,n,j] = sum_{m in slice s} w[n,m] R[m,j]  each thread owns a float4 column, R rows
//                streamed coalesced; m is split in slices for occupancy, slices reduced in fixed order
//   k_finalize   neg = sum_s part / den ; x <- x - scale*neg (RBF) | x + scale*(xq*sum_w - sum_s part) (SPARSE)
//
// Direct differences are used for the distance (not |x|^2+|r|^2-2x.r): same traffic, no cancellation.
// Results are deterministic (no float atomics).
#include "sdn_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kQ = 8;          // queries handled per workgroup (register accumulators)
constexpr int kMaxSlices = 32;

struct Plan {
  int n_chunks, d_tiles, slices, m_per_slice;
  size_t off_xq, off_d2, off_w, off_den, off_part, total;
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

inline Plan make_plan(int N, int M, int64_t D) {
  Plan p;
  p.n_chunks = (N + kQ - 1) / kQ;
  p.d_tiles = (int)((D / 4 + kThreads - 1) / kThreads);
  int want = 1024 / (p.d_tiles * p.n_chunks > 0 ? p.d_tiles * p.n_chunks : 1);   // ~4 workgroups per CU
  if (want < 1) want = 1;
  if (want > kMaxSlices) want = kMaxSlices;
  if (want > M) want = M > 0 ? M : 1;
  p.m_per_slice = (M + want - 1) / want;
  if (p.m_per_slice < 1) p.m_per_slice = 1;
  if (p.m_per_slice > 1024) p.m_per_slice = 1024;   // k_wsum stages kQ*m_per_slice weights in LDS (<= 32 KiB)
  p.slices = (M + p.m_per_slice - 1) / p.m_per_slice;
  if (p.slices < 1) p.slices = 1;
  size_t o = 0;
  p.off_xq = o;   o += align256((size_t)N * D * 4);
  p.off_d2 = o;   o += align256((size_t)N * M * 4);
  p.off_w = o;    o += align256((size_t)N * M * 4);
  p.off_den = o;  o += align256((size_t)N * 4);
  p.off_part = o; o += align256((size_t)p.slices * N * D * 4);
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

// ---- squared distances ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
k_dist2(const float* __restrict__ xq, const float* __restrict__ R, int N, int M, int64_t D,
        float* __restrict__ d2) {
  __shared__ float red[kQ][4];
  const int m = blockIdx.x;
  const int n0 = blockIdx.y * kQ;
  const int nq = min(kQ, N - n0);
  const float4* r4 = reinterpret_cast<const float4*>(R + (int64_t)m * D);
  const int64_t d4 = D / 4;
  float acc[kQ];
#pragma unroll
  for (int q = 0; q < kQ; ++q) acc[q] = 0.f;
  for (int64_t j = threadIdx.x; j < d4; j += kThreads) {
    const float4 r = r4[j];
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      if (q < nq) {
        const float4 a = reinterpret_cast<const float4*>(xq + (int64_t)(n0 + q) * D)[j];
        const float e0 = a.x - r.x, e1 = a.y - r.y, e2 = a.z - r.z, e3 = a.w - r.w;
        acc[q] = fmaf(e0, e0, fmaf(e1, e1, fmaf(e2, e2, fmaf(e3, e3, acc[q]))));
      }
    }
  }
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < kQ; ++q) {
    const float s = wave_sum(acc[q]);
    if (lane == 0) red[q][wid] = s;
  }
  __syncthreads();
  if (threadIdx.x < nq) {
    const int q = threadIdx.x;
    d2[(int64_t)(n0 + q) * M + m] = (red[q][0] + red[q][1]) + (red[q][2] + red[q][3]);
  }
}

// ---- squared distances, batched form: a workgroup owns an 8-ref x 8-query tile, so every reference row fetched is
//      used for 8 queries and every query row for 8 references (the row-per-workgroup form re-reads the N query rows
//      once per reference: 1.8 GB of L2 fills per call at N = 32, measured with FETCH_SIZE) ------------------------
constexpr int kR = 8;
__global__ void __launch_bounds__(kThreads)
k_dist2_tile(const float* __restrict__ xq, const float* __restrict__ R, int N, int M, int64_t D,
             float* __restrict__ d2) {
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

// ---- weights, denominator, gate -----------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
k_weights(const float* __restrict__ d2, int M, int weight_fn, float inv_two_sigma_sq, float radius, float eps,
          float gate, float* __restrict__ w, float* __restrict__ den_ws, float* __restrict__ out_den,
          int32_t* __restrict__ out_isneg) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  float s = 0.f;
  for (int m = threadIdx.x; m < M; m += kThreads) {
    const float dist = sqrtf(d2[(int64_t)n * M + m]);
    float wv;
    if (weight_fn == SDN_REPEL_RBF) {
      wv = expf(-dist * inv_two_sigma_sq);
    } else {
      wv = (dist < radius) ? fmaxf(radius / dist - 1.f, 0.f) : 0.f;   // dist NaN -> not a neighbour
    }
    w[(int64_t)n * M + m] = wv;
    s += wv;
  }
  const float tot = block_sum<4>(s, red);
  if (threadIdx.x == 0) {
    const float den = (weight_fn == SDN_REPEL_RBF) ? tot + eps : tot;
    den_ws[n] = den;
    if (out_den) out_den[n] = den;
    if (out_isneg) out_isneg[n] = (weight_fn == SDN_REPEL_RBF) ? (den > gate ? 1 : 0) : (tot != 0.f ? 1 : 0);
  }
}

// ---- weighted sum of reference rows, one m-slice per blockIdx.z ----------------------------------
__global__ void __launch_bounds__(kThreads)
k_wsum(const float* __restrict__ w, const float* __restrict__ R, int N, int M, int64_t D, int m_per_slice,
       float* __restrict__ part) {
  extern __shared__ float wl[];                  // [kQ][m_per_slice]
  const int n0 = blockIdx.y * kQ;
  const int nq = min(kQ, N - n0);
  const int m_lo = blockIdx.z * m_per_slice;
  const int m_hi = min(M, m_lo + m_per_slice);
  const int mc = m_hi - m_lo;
  for (int i = threadIdx.x; i < kQ * m_per_slice; i += kThreads) {
    const int q = i / m_per_slice, mm = i - q * m_per_slice;
    wl[i] = (q < nq && mm < mc) ? w[(int64_t)(n0 + q) * M + m_lo + mm] : 0.f;
  }
  __syncthreads();
  const int64_t d4 = D / 4;
  const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (j >= d4) return;
  float4 acc[kQ];
#pragma unroll
  for (int q = 0; q < kQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* r4 = reinterpret_cast<const float4*>(R) + j;
#pragma unroll 4
  for (int mm = 0; mm < mc; ++mm) {
    const float4 r = r4[(int64_t)(m_lo + mm) * d4];
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      const float wv = wl[q * m_per_slice + mm];
      acc[q].x = fmaf(wv, r.x, acc[q].x); acc[q].y = fmaf(wv, r.y, acc[q].y);
      acc[q].z = fmaf(wv, r.z, acc[q].z); acc[q].w = fmaf(wv, r.w, acc[q].w);
    }
  }
  float4* po = reinterpret_cast<float4*>(part) + ((int64_t)blockIdx.z * N + n0) * d4 + j;
#pragma unroll
  for (int q = 0; q < kQ; ++q)
    if (q < nq) po[(int64_t)q * d4] = acc[q];
}

// ---- reduce slices + epilogue --------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
k_finalize(float* __restrict__ x, const float* __restrict__ xq, const float* __restrict__ part,
           const float* __restrict__ den_ws, int N, int64_t D, int slices, int weight_fn, float scale,
           float* __restrict__ out_neg) {
  const int n = blockIdx.y;
  const int64_t d4 = D / 4;
  const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (j >= d4) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < slices; ++z) {
    const float4 p = reinterpret_cast<const float4*>(part)[((int64_t)z * N + n) * d4 + j];
    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
  }
  const float den = den_ws[n];
  float4* xp = reinterpret_cast<float4*>(x) + (int64_t)n * d4 + j;
  float4 xv = *xp, g;
  if (weight_fn == SDN_REPEL_RBF) {
    g.x = s.x / den; g.y = s.y / den; g.z = s.z / den; g.w = s.w / den;
    xv.x -= scale * g.x; xv.y -= scale * g.y; xv.z -= scale * g.z; xv.w -= scale * g.w;
  } else {
    const float4 q = reinterpret_cast<const float4*>(xq)[(int64_t)n * d4 + j];
    g.x = q.x * den - s.x; g.y = q.y * den - s.y; g.z = q.z * den - s.z; g.w = q.w * den - s.w;
    xv.x += scale * g.x; xv.y += scale * g.y; xv.z += scale * g.z; xv.w += scale * g.w;
  }
  *xp = xv;
  if (out_neg) reinterpret_cast<float4*>(out_neg)[(int64_t)n * d4 + j] = g;
}

// ---- calibration tails ----------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
k_beta_rows(const float* __restrict__ d2, int M, float inv_two_sigma_sq, float eps, float* __restrict__ beta) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  float s = 0.f;
  for (int m = threadIdx.x; m < M; m += kThreads) s += expf(-sqrtf(d2[(int64_t)n * M + m]) * inv_two_sigma_sq);
  const float tot = block_sum<4>(s, red);
  if (threadIdx.x == 0) beta[n] = tot + eps;
}

__global__ void __launch_bounds__(kThreads)
k_sqrt_rows(const float* __restrict__ d2, int64_t total, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads)
    out[i] = sqrtf(d2[i]);
}

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

// Shared front half: (qnorm) + distances.  Returns the pointer the distances were computed from.
inline const float* run_dist(const sdn_repel_params* p, const Plan& pl, const float* x, const float* R, char* ws,
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
  if (N >= 4)
    hipLaunchKernelGGL(k_dist2_tile, dim3((M + kR - 1) / kR, pl.n_chunks), dim3(kThreads), 0, st, xq, R, N, M, D,
                       reinterpret_cast<float*>(ws + pl.off_d2));
  else
    hipLaunchKernelGGL(k_dist2, dim3(M, pl.n_chunks), dim3(kThreads), 0, st, xq, R, N, M, D,
                       reinterpret_cast<float*>(ws + pl.off_d2));
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
  float* part = reinterpret_cast<float*>(ws + pl.off_part);

  const float* xq = x;
  int slices = pl.slices;
  if (M > 0) {
    xq = run_dist(p, pl, x, R, ws, st);
  } else {
    slices = 0;                                   // empty reference set: den = eps (RBF) / 0 (SPARSE), neg = 0
    if (p->qnorm == SDN_QNORM_CHANNEL) {
      float* q = reinterpret_cast<float*>(ws + pl.off_xq);
      hipLaunchKernelGGL(k_qnorm, dim3((p->hw + kThreads - 1) / kThreads, N), dim3(kThreads), 0, st, x, q,
                         p->channels, p->hw);
      xq = q;
    }
  }
  hipLaunchKernelGGL(k_weights, dim3(N), dim3(kThreads), 0, st, reinterpret_cast<const float*>(ws + pl.off_d2), M,
                     p->weight_fn, 1.f / (2.f * p->sigma * p->sigma), p->radius, p->epsilon, p->gate, w, den,
                     out_den, out_isneg);
  if (M > 0) {
    const size_t lds = (size_t)kQ * pl.m_per_slice * sizeof(float);
    hipLaunchKernelGGL(k_wsum, dim3(pl.d_tiles, pl.n_chunks, pl.slices), dim3(kThreads), lds, st, w, R, N, M, D,
                       pl.m_per_slice, part);
  }
  hipLaunchKernelGGL(k_finalize, dim3(pl.d_tiles, N), dim3(kThreads), 0, st, x, xq, part, den, N, D, slices,
                     p->weight_fn, p->scale, out_neg);
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
  run_dist(p, pl, queries, R, ws, st);
  const float* d2 = reinterpret_cast<const float*>(ws + pl.off_d2);
  if (p->weight_fn == SDN_REPEL_RBF) {
    hipLaunchKernelGGL(k_beta_rows, dim3(N), dim3(kThreads), 0, st, d2, M, 1.f / (2.f * p->sigma * p->sigma),
                       p->epsilon, out);
  } else {
    const int64_t total = (int64_t)N * M;
    int g = (int)((total + kThreads - 1) / kThreads);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_sqrt_rows, dim3(g), dim3(kThreads), 0, st, d2, total, out);
  }
  return sdn_launch_status();
}

}  // extern "C"
