// GroupNorm(+SiLU), LayerNorm, conv_in and the timestep embedding: the HBM-bound operators of the UNet
// (rows U1, U3-U5).  All read bf16 NHWC in 16-byte chunks, accumulate in fp32 and write bf16 once.
#include "sdn_common.h"
#include "sdn_ops.h"

namespace sdn_norm_detail {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int THREADS = 256;
constexpr int GN_MAX_TILES = 128;

template <typename T>
__device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
  f[0] = T::to_f(v.x & 0xffff); f[1] = T::to_f(v.x >> 16); f[2] = T::to_f(v.y & 0xffff); f[3] = T::to_f(v.y >> 16);
  f[4] = T::to_f(v.z & 0xffff); f[5] = T::to_f(v.z >> 16); f[6] = T::to_f(v.w & 0xffff); f[7] = T::to_f(v.w >> 16);
}
template <typename T>
__device__ __forceinline__ u32x4 pack8(const float* f) {
  return (u32x4){T::pack2(f[0], f[1]), T::pack2(f[2], f[3]), T::pack2(f[4], f[5]), T::pack2(f[6], f[7])};
}
__device__ __forceinline__ float silu_f(float v) { return v / (1.f + __expf(-v)); }

// ------------------------------------------------------------------------------------------------
// GroupNorm pass 1: per (sample, row tile) partial sums per group, deterministic.
// Thread (cl, rl): cl owns NCH fixed 8-channel chunks, rl strides over the tile's rows.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(THREADS)
k_gn_stats(const unsigned short* __restrict__ x, const unsigned short* __restrict__ x2, int hw, int c1, int c2,
           int groups, int rows_per_tile, int ct, int nch, float* __restrict__ partials) {
  extern __shared__ float lds[];                 // [rt][C][2]
  const int C = c1 + c2, cpg = C / groups;
  const int b = blockIdx.x, tile = blockIdx.y, ntiles = gridDim.y;
  const int rt = THREADS / ct;
  const int cl = threadIdx.x % ct, rl = threadIdx.x / ct;
  const int r_lo = tile * rows_per_tile, r_hi = min(hw, r_lo + rows_per_tile);
  if (rl < rt) {
    for (int q = 0; q < nch; ++q) {
      const int ch0 = (cl + q * ct) * 8;
      const unsigned short* src; int ld, cc;
      if (ch0 < c1) { src = x; ld = c1; cc = ch0; } else { src = x2; ld = c2; cc = ch0 - c1; }
      float s[8], ss[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
      // 4 rows in flight per thread: the loop is latency-bound otherwise (one dependent 16-B load per iteration)
      for (int r = r_lo + rl; r < r_hi; r += 4 * rt) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rr = min(r + j * rt, r_hi - 1);
          v[j] = *reinterpret_cast<const u32x4*>(src + ((long)b * hw + rr) * ld + cc);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (r + j * rt < r_hi) {
            float f[8];
            unpack8<T>(v[j], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] = fmaf(f[e], f[e], ss[e]); }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        lds[((long)rl * C + ch0 + e) * 2 + 0] = s[e];
        lds[((long)rl * C + ch0 + e) * 2 + 1] = ss[e];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < groups) {
    const int gidx = threadIdx.x;
    float s = 0.f, ss = 0.f;
    for (int r = 0; r < rt; ++r)
      for (int c = gidx * cpg; c < (gidx + 1) * cpg; ++c) {
        s += lds[((long)r * C + c) * 2 + 0];
        ss += lds[((long)r * C + c) * 2 + 1];
      }
    float* o = partials + (((long)b * ntiles + tile) * groups + gidx) * 2;
    o[0] = s; o[1] = ss;
  }
}

// GroupNorm pass 1b: reduce the row-tile partials -> mean / rstd per (sample, group).  One workgroup per sample;
// 8 threads per group each sum a strided share of the tiles (loads in parallel, not a 128-deep dependent chain),
// then lane 0 of each octet adds the 8 shares in fixed order (deterministic).
__global__ void __launch_bounds__(512)
k_gn_finalize(const float* __restrict__ partials, int ntiles, int groups, float n, float eps,
              float* __restrict__ stats) {
  __shared__ float sh[64][8][2];
  const int b = blockIdx.x, gi = threadIdx.x >> 3, part = threadIdx.x & 7;
  if (gi < groups) {
    float s = 0.f, ss = 0.f;
    for (int t = part; t < ntiles; t += 8) {
      const float* p = partials + (((long)b * ntiles + t) * groups + gi) * 2;
      s += p[0]; ss += p[1];
    }
    sh[gi][part][0] = s; sh[gi][part][1] = ss;
  }
  __syncthreads();
  if (gi < groups && part == 0) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += sh[gi][k][0]; ss += sh[gi][k][1]; }
    const float mean = s / n;
    const float var = fmaxf(ss / n - mean * mean, 0.f);
    stats[((long)b * groups + gi) * 2 + 0] = mean;
    stats[((long)b * groups + gi) * 2 + 1] = rsqrtf(var + eps);
  }
}

// Statistics from the per-column partials the producing GEMMs left behind (sdn_gemm_stats_*): cols [hw/128 blocks of the
// sample][c][2] = (sum, sum of squares) per 128-row block.  One workgroup per (sample, group): its threads stride over the
// group's channels x the sample's blocks (2048 blocks at 512 x 512), then a fixed-order workgroup reduction.  Same output
// as k_gn_finalize; the summation order depends on neither the batch size nor the producer's tile.
__global__ void __launch_bounds__(256)
k_gn_finalize_cols(const float* __restrict__ cols1, const float* __restrict__ cols2, int blocks_per_sample, int c1, int c2,
                   int groups, float n, float eps, float* __restrict__ stats) {
  __shared__ float red[8];
  const int C = c1 + c2, cpg = C / groups, b = blockIdx.x, gi = blockIdx.y;
  const int total = blocks_per_sample * cpg;
  float s = 0.f, q = 0.f;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int t = e / cpg, c = gi * cpg + (e - t * cpg);
    const bool second = c >= c1;
    const float* src = second ? cols2 : cols1;
    const int cw = second ? c2 : c1, cc = second ? c - c1 : c;
    const float2 v = *reinterpret_cast<const float2*>(src + 2 * (((long)b * blocks_per_sample + t) * cw + cc));
    s += v.x; q += v.y;
  }
  s = block_sum<4>(s, red);
  q = block_sum<4>(q, red + 4);
  if (threadIdx.x == 0) {
    const float mean = s / n;
    const float var = fmaxf(q / n - mean * mean, 0.f);
    stats[((long)b * groups + gi) * 2 + 0] = mean;
    stats[((long)b * groups + gi) * 2 + 1] = rsqrtf(var + eps);
  }
}

// GroupNorm pass 2: normalise + affine (+SiLU), write the (concatenated) map.
template <typename T>
__global__ void __launch_bounds__(THREADS)
k_gn_apply(const unsigned short* __restrict__ x, const unsigned short* __restrict__ x2, int hw, int c1, int c2,
           int groups, int rows_per_block, int silu, const float* __restrict__ gamma,
           const float* __restrict__ beta, const float* __restrict__ stats, unsigned short* __restrict__ out) {
  // per-channel affine of this sample, built once per workgroup: y = x * ca[c] + cb[c]   (ca = rstd_g * gamma_c,
  // cb = beta_c - mean_g * ca) -- the channel -> group division leaves the per-element loop
  extern __shared__ float aff[];                 // [2][C]
  const int C = c1 + c2, cpg = C / groups, cchunks = C / 8;
  const int b = blockIdx.x;
  float* ca = aff; float* cb = aff + C;
  for (int c = threadIdx.x; c < C; c += THREADS) {
    const int gi = c / cpg;
    const float mean = stats[((long)b * groups + gi) * 2 + 0], rstd = stats[((long)b * groups + gi) * 2 + 1];
    const float a = rstd * gamma[c];
    ca[c] = a; cb[c] = beta[c] - mean * a;
  }
  __syncthreads();
  const int r_lo = blockIdx.y * rows_per_block, r_hi = min(hw, r_lo + rows_per_block);
  const long total = (long)(r_hi - r_lo) * cchunks;
  for (long e = threadIdx.x; e < total; e += THREADS) {
    const int r = r_lo + (int)(e / cchunks), ch0 = (int)(e % cchunks) * 8;
    const unsigned short* src; int ld, cc;
    if (ch0 < c1) { src = x; ld = c1; cc = ch0; } else { src = x2; ld = c2; cc = ch0 - c1; }
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + ((long)b * hw + r) * ld + cc);
    float f[8];
    unpack8<T>(v, f);
    const float4 a0 = *reinterpret_cast<const float4*>(ca + ch0), a1 = *reinterpret_cast<const float4*>(ca + ch0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(cb + ch0), b1 = *reinterpret_cast<const float4*>(cb + ch0 + 4);
    const float am[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float bm[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float y = fmaf(f[k], am[k], bm[k]);
      f[k] = silu ? silu_f(y) : y;
    }
    *reinterpret_cast<u32x4*>(out + ((long)b * hw + r) * C + ch0) = pack8<T>(f);
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave normalises R consecutive rows, each held in registers as NQ 16-byte chunks per lane
// (C <= 512 * NQ); two-pass variance.  All R * NQ loads of a wave are issued before the first reduction: with one
// 640-byte row per wave (R = 1) the kernel ran at 2.8 TB/s, bound by wave turnover rather than by HBM.
// ------------------------------------------------------------------------------------------------
template <typename T, int NQ, int R>
__global__ void __launch_bounds__(THREADS)
k_layernorm(const unsigned short* __restrict__ x, long rows, int C, float eps, const float* __restrict__ gamma,
            const float* __restrict__ beta, unsigned short* __restrict__ out, int mod, int rows_per_batch, int ld_mod) {
  // mod = 0: y = LN(x) * gamma[c] + beta[c]            (BasicTransformerBlock norms)
  // mod = 1: y = LN(x) * (1 + gamma[b, c]) + beta[b, c] (adaLN: gamma = scale, beta = shift, per-sample rows of ld_mod)
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  const int cchunks = C / 8;
  float f[R][NQ][8];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = 0.f;
    const long row = row0 + r < rows ? row0 + r : rows - 1;          // clamped: tail rows recompute the last row, never store it
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int cc = lane + q * 64;
      if (cc < cchunks) {
        unpack8<T>(*reinterpret_cast<const u32x4*>(x + row * C + cc * 8), f[r][q]);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[r] += f[r][q][k];
      }
    }
  }
  float mean[R], rstd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) mean[r] = wave_sum(s[r]) / (float)C;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int cc = lane + q * 64;
      if (cc < cchunks) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = f[r][q][k] - mean[r]; ss = fmaf(d, d, ss); }
      }
    }
    s[r] = ss;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(wave_sum(s[r]) / (float)C + eps);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int cc = lane + q * 64;
    if (cc < cchunks) {
      float gm[8], bt[8];
      auto load_affine = [&](const float* gp, const float* bp) {
        const float4 g0 = *reinterpret_cast<const float4*>(gp + cc * 8), g1 = *reinterpret_cast<const float4*>(gp + cc * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(bp + cc * 8), b1 = *reinterpret_cast<const float4*>(bp + cc * 8 + 4);
        gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w; gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
        bt[0] = b0.x; bt[1] = b0.y; bt[2] = b0.z; bt[3] = b0.w; bt[4] = b1.x; bt[5] = b1.y; bt[6] = b1.z; bt[7] = b1.w;
      };
      if (!mod) load_affine(gamma, beta);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const long row = row0 + r;
        if (row < rows) {
          if (mod) {
            const long b = row / rows_per_batch;
            load_affine(gamma + b * ld_mod, beta + b * ld_mod);
          }
          float y[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) y[k] = (f[r][q][k] - mean[r]) * rstd[r] * (mod ? 1.f + gm[k] : gm[k]) + bt[k];
          *reinterpret_cast<u32x4*>(out + row * C + cc * 8) = pack8<T>(y);
        }
      }
    }
  }
}

// Row statistics only (mean, rstd) for the LayerNorm-folded GEMM with wide N: a read-only pass, two-pass variance from
// registers like k_layernorm.  out [rows][2] f32.
template <typename T, int NQ, int R>
__global__ void __launch_bounds__(THREADS)
k_row_stats(const unsigned short* __restrict__ x, long rows, int C, float eps, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  const int cchunks = C / 8;
  float f[R][NQ][8];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = 0.f;
    const long row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int cc = lane + q * 64;
      if (cc < cchunks) {
        unpack8<T>(*reinterpret_cast<const u32x4*>(x + row * C + cc * 8), f[r][q]);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[r] += f[r][q][k];
      }
    }
  }
  float mean[R];
#pragma unroll
  for (int r = 0; r < R; ++r) mean[r] = wave_sum(s[r]) / (float)C;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int cc = lane + q * 64;
      if (cc < cchunks) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = f[r][q][k] - mean[r]; ss = fmaf(d, d, ss); }
      }
    }
    s[r] = ss;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float rstd = rsqrtf(wave_sum(s[r]) / (float)C + eps);
    if (lane == 0 && row0 + r < rows) *reinterpret_cast<float2*>(out + 2 * (row0 + r)) = make_float2(mean[r], rstd);
  }
}

// ------------------------------------------------------------------------------------------------
// conv_in: direct 3x3 conv, fp32 NCHW latent -> NHWC bf16.  One thread = one pixel x 8 output channels.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(THREADS)
k_conv_in(const float* __restrict__ lat, const unsigned short* __restrict__ w, const float* __restrict__ bias,
          int B, int cin, int H, int W, int cout, int ldo, unsigned short* __restrict__ out) {
  // cout = the channels THIS launch computes (w / bias / out already point at its first channel); ldo = channels per pixel
  extern __shared__ float wl[];                   // [9*cin][2 halves][cout/8 chunks][4] as f32: consecutive lanes
  const int kk = 9 * cin;                         // (channel chunks) read consecutive 16-B slots -> conflict-free b128
  for (int i = threadIdx.x; i < cout * kk; i += THREADS) {
    const int co = i / kk, k = i - co * kk;
    wl[((k * 2 + ((co >> 2) & 1)) * (cout / 8) + (co >> 3)) * 4 + (co & 3)] = T::to_f(w[i]);
  }
  __syncthreads();
  // One thread = PX = 4 horizontally adjacent pixels x 8 output channels: the 3 x 6 input window and every weight read
  // from LDS are reused for the four pixels (index math, global loads and LDS reads per FMA drop 4x); lanes run over the
  // channel chunks of a pixel group, so a wave's stores are whole contiguous rows.  32-bit index math throughout (the
  // host checks the range): the 64-bit divisions this loop used to do per output chunk cost more than its FMAs.
  constexpr int PX = 4;
  const int cchunks = cout / 8;
  const int WQ = (W + PX - 1) / PX;
  const int total = B * H * WQ * cchunks;
  const int HWQ = H * WQ;
  for (int e = blockIdx.x * THREADS + threadIdx.x; e < total; e += gridDim.x * THREADS) {
    const int pq = e / cchunks, cc = e - pq * cchunks;
    const int b = pq / HWQ, rem = pq - b * HWQ;
    const int yh = rem / WQ, x0 = (rem - yh * WQ) * PX;
    float acc[PX][8];
    {
      const float4 b0 = *reinterpret_cast<const float4*>(bias + cc * 8), b1 = *reinterpret_cast<const float4*>(bias + cc * 8 + 4);
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        acc[p][0] = b0.x; acc[p][1] = b0.y; acc[p][2] = b0.z; acc[p][3] = b0.w;
        acc[p][4] = b1.x; acc[p][5] = b1.y; acc[p][6] = b1.z; acc[p][7] = b1.w;
      }
    }
    const float* lb = lat + (long)b * cin * H * W;
    for (int c = 0; c < cin; ++c) {
      float v[3][PX + 2];                         // rows yh-1..yh+1, columns x0-1..x0+PX: clamped loads, zeroed by select
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < PX + 2; ++tx) {
          const int iy = yh + ty - 1, ix = x0 + tx - 1;
          const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
          const float t = lb[((long)c * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)];
          v[ty][tx] = ok ? t : 0.f;
        }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float4 w0 = *reinterpret_cast<const float4*>(&wl[(((tap * cin + c) * 2 + 0) * cchunks + cc) * 4]);
        const float4 w1 = *reinterpret_cast<const float4*>(&wl[(((tap * cin + c) * 2 + 1) * cchunks + cc) * 4]);
#pragma unroll
        for (int p = 0; p < PX; ++p) {
          const float x = v[tap / 3][tap % 3 + p];
          acc[p][0] = fmaf(x, w0.x, acc[p][0]); acc[p][1] = fmaf(x, w0.y, acc[p][1]);
          acc[p][2] = fmaf(x, w0.z, acc[p][2]); acc[p][3] = fmaf(x, w0.w, acc[p][3]);
          acc[p][4] = fmaf(x, w1.x, acc[p][4]); acc[p][5] = fmaf(x, w1.y, acc[p][5]);
          acc[p][6] = fmaf(x, w1.z, acc[p][6]); acc[p][7] = fmaf(x, w1.w, acc[p][7]);
        }
      }
    }
    const long pix0 = ((long)b * H + yh) * W + x0;
#pragma unroll
    for (int p = 0; p < PX; ++p)
      if (x0 + p < W) *reinterpret_cast<u32x4*>(out + (pix0 + p) * ldo + cc * 8) = pack8<T>(acc[p]);
  }
}

template <typename T>
__global__ void k_temb(float t_val, const float* __restrict__ t_dev, int B, int dim, unsigned short* __restrict__ out) {
  const float t = t_dev ? *t_dev : t_val;           // t_dev: graph replays read the step's timestep from device memory
  const int half = dim / 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * dim; i += gridDim.x * blockDim.x) {
    const int k = i % dim;
    const int kk = k < half ? k : k - half;
    const float f = expf(-9.210340371976184f * (float)kk / (float)half);   // ln(10000)
    const float a = t * f;
    const float v = k < half ? cosf(a) : sinf(a);
    out[i] = (unsigned short)(T::pack2(v, 0.f) & 0xffff);
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }


template <typename T>
int groupnorm_impl(const void* x, const void* x2, int32_t batch, int32_t hw, int32_t c1, int32_t c2, int32_t groups,
                   float eps, int32_t silu, const float* gamma, const float* beta, void* out, float* stats_ws,
                   void* stream, const float* cols1 = nullptr, const float* cols2 = nullptr) {
  if (!x || !gamma || !beta || !out || !stats_ws || batch < 0 || hw <= 0 || c1 <= 0 || c2 < 0 || groups <= 0 ||
      groups > 64)
    return SDN_E_INVALID;
  if (c2 > 0 && !x2) return SDN_E_INVALID;
  const int C = c1 + c2;
  if ((c1 & 7) || (c2 & 7) || C % groups != 0 || C > 4096) return SDN_E_INVALID;
  if (!al16(x) || (x2 && !al16(x2)) || !al16(out) || !al16(gamma) || !al16(beta)) return SDN_E_INVALID;
  if (batch == 0) return SDN_OK;
  const int cch = C / 8;
  int nch = (cch + THREADS - 1) / THREADS;
  while (cch % nch != 0) ++nch;
  const int ct = cch / nch;
  if (ct > THREADS) return SDN_E_INVALID;
  const int rt = THREADS / ct;
  int ntiles = (hw + 31) / 32;                // 32 rows per stats tile: thousands of workgroups at the 64x64 level
  if (ntiles > GN_MAX_TILES) ntiles = GN_MAX_TILES;
  const int rows_per_tile = (hw + ntiles - 1) / ntiles;
  ntiles = (hw + rows_per_tile - 1) / rows_per_tile;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)rt * C * 2 * sizeof(float);
  if (lds > 64 * 1024) return SDN_E_INVALID;
  if (cols1) {                                // statistics from the producers' column partials: no pass over x
    if ((hw & 127) || (c2 > 0 && !cols2)) return SDN_E_INVALID;
    hipLaunchKernelGGL(k_gn_finalize_cols, dim3(batch, groups), dim3(256), 0, st, cols1, cols2, hw / 128, c1, c2, groups,
                       (float)hw * (float)(C / groups), eps, stats_ws);
  } else {
    float* partials = stats_ws + (size_t)batch * groups * 2;          // [B][ntiles][G][2] after the final stats
    hipLaunchKernelGGL((k_gn_stats<T>), dim3(batch, ntiles), dim3(THREADS), lds, st, (const unsigned short*)x,
                       (const unsigned short*)x2, hw, c1, c2, groups, rows_per_tile, ct, nch, partials);
    hipLaunchKernelGGL(k_gn_finalize, dim3(batch), dim3(512), 0, st, partials, ntiles, groups,
                       (float)hw * (float)(C / groups), eps, stats_ws);
  }
  int rows_per_block = (256 * 8 * 16) / C;    // ~16 chunks per thread (amortises the per-workgroup affine table)
  if (rows_per_block < 1) rows_per_block = 1;
  const int nblk = (hw + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL((k_gn_apply<T>), dim3(batch, nblk), dim3(THREADS), (size_t)2 * C * sizeof(float), st, (const unsigned short*)x,
                     (const unsigned short*)x2, hw, c1, c2, groups, rows_per_block, silu, gamma, beta, stats_ws,
                     (unsigned short*)out);
  return sdn_launch_status();
}

template <typename T>
int layernorm_impl(const void* x, int64_t rows, int32_t c, float eps, const float* gamma, const float* beta, void* out,
                   void* stream, int mod = 0, int rows_per_batch = 0, int ld_mod = 0) {
  if (!x || !gamma || !beta || !out || rows < 0 || c <= 0 || (c & 7) || c > 2048) return SDN_E_INVALID;
  if (!al16(x) || !al16(out) || !al16(gamma) || !al16(beta)) return SDN_E_INVALID;
  if (mod && (rows_per_batch <= 0 || ld_mod < c || (ld_mod & 3))) return SDN_E_INVALID;
  if (rows == 0) return SDN_OK;
#define SDN_LN_LAUNCH(NQ, R)                                                                                          \
  hipLaunchKernelGGL((k_layernorm<T, NQ, R>), dim3((unsigned)((rows + 4 * (R) - 1) / (4 * (R)))), dim3(THREADS), 0,      \
                     (hipStream_t)stream, (const unsigned short*)x, (long)rows, c, eps, gamma, beta, (unsigned short*)out, \
                     mod, rows_per_batch, ld_mod)
  if (c <= 512) SDN_LN_LAUNCH(1, 4);
  else if (c <= 1024) SDN_LN_LAUNCH(2, 2);
  else SDN_LN_LAUNCH(4, 1);
#undef SDN_LN_LAUNCH
  return sdn_launch_status();
}

// ---- MMDiT patch embedding front / back ends ----------------------------------------------------------
// patchify: fp32 NCHW latent [B,C,H,W] -> 16-bit [B*(H/p)*(W/p), C*p*p], column order (c, py, px) = the flattened
// conv weight [O, C, p, p] of PatchEmbed.proj, so the patch embedding is one GEMM (K = C*p*p = 64 for SD-v3).
template <typename T>
__global__ void __launch_bounds__(THREADS)
k_patchify(const float* __restrict__ lat, int B, int C, int H, int W, int p, unsigned short* __restrict__ out) {
  const int hp = H / p, wp = W / p, K = C * p * p;
  const long total = (long)B * hp * wp * K;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += (long)gridDim.x * THREADS) {
    const int k = (int)(e % K);
    const long tok = e / K;
    const int px = k % p, py = (k / p) % p, c = k / (p * p);
    const int tx = (int)(tok % wp), ty = (int)((tok / wp) % hp), b = (int)(tok / ((long)wp * hp));
    const float v = lat[(((long)b * C + c) * H + ty * p + py) * W + tx * p + px];
    out[e] = (unsigned short)(T::pack2(v, 0.f) & 0xffff);
  }
}
// unpatchify: fp32 tokens [B*hp*wp, p*p*C] (column = (py*p + px)*C + c, diffusers' "nhwpqc->nchpwq") -> fp32 NCHW.
__global__ void __launch_bounds__(THREADS)
k_unpatchify(const float* __restrict__ tok, int B, int C, int H, int W, int p, float* __restrict__ out) {
  const int hp = H / p, wp = W / p;
  const long total = (long)B * C * H * W;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += (long)gridDim.x * THREADS) {
    const int x = (int)(e % W), y = (int)((e / W) % H), c = (int)((e / ((long)W * H)) % C), b = (int)(e / ((long)W * H * C));
    const long t = ((long)b * hp + y / p) * wp + x / p;
    out[e] = tok[t * (p * p * C) + ((y % p) * p + (x % p)) * C + c];
  }
}
template <typename T>
int patchify_impl(const float* lat, int32_t B, int32_t C, int32_t H, int32_t W, int32_t p, void* out, void* stream) {
  if (!lat || !out || B < 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0 || H % p || W % p) return SDN_E_INVALID;
  if (B == 0) return SDN_OK;
  const long total = (long)B * C * H * W;
  long grid = (total + THREADS - 1) / THREADS;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL((k_patchify<T>), dim3((unsigned)grid), dim3(THREADS), 0, (hipStream_t)stream, lat, B, C, H, W, p,
                     (unsigned short*)out);
  return sdn_launch_status();
}

template <typename T>
int conv_in_impl(const float* lat, const void* w, const float* bias, int32_t batch, int32_t cin, int32_t h, int32_t wd,
                 int32_t cout, void* out, void* stream) {
  if (!lat || !w || !bias || !out || batch < 0 || cin <= 0 || cin > 16 || h <= 0 || wd <= 0 || cout <= 0 ||
      (cout & 7) || !al16(out))
    return SDN_E_INVALID;
  if (batch == 0) return SDN_OK;
  // the weights of one launch live in LDS as f32 (<= 64 KB): wide layers (VAE: 4 -> 512, SD-v3 VAE: 16 -> 512) are cut
  // into channel chunks, each launch writing its slice of the NHWC rows
  int chunk = (int)((64 * 1024) / ((size_t)9 * cin * sizeof(float))) & ~7;
  if (chunk < 8) return SDN_E_INVALID;
  if (chunk > cout) chunk = cout;
  if ((long)batch * h * wd * (chunk / 8) >= (1L << 31)) return SDN_E_INVALID;
  for (int co0 = 0; co0 < cout; co0 += chunk) {
    const int cc = cout - co0 < chunk ? cout - co0 : chunk;
    const size_t lds = (size_t)cc * 9 * cin * sizeof(float);
    const long total = (long)batch * h * ((wd + 3) / 4) * (cc / 8);
    long grid = (total + THREADS - 1) / THREADS;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL((k_conv_in<T>), dim3((unsigned)grid), dim3(THREADS), lds, (hipStream_t)stream, lat,
                       (const unsigned short*)w + (size_t)co0 * 9 * cin, bias + co0, batch, cin, h, wd, cc, cout,
                       (unsigned short*)out + co0);
  }
  return sdn_launch_status();
}

template <typename T>
int temb_impl(float timestep, int32_t batch, int32_t dim, void* out, void* stream, const float* t_dev = nullptr) {
  if (!out || batch < 0 || dim <= 0 || (dim & 1)) return SDN_E_INVALID;
  if (batch == 0) return SDN_OK;
  const int n = batch * dim;
  hipLaunchKernelGGL((k_temb<T>), dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, timestep, t_dev, batch, dim,
                     (unsigned short*)out);
  return sdn_launch_status();
}

template <typename T>
int row_stats_impl(const void* x, int64_t rows, int32_t c, float eps, float* out, void* stream) {
  if (!x || !out || rows < 0 || c <= 0 || (c & 7) || c > 2048 || !al16(x) || (reinterpret_cast<uintptr_t>(out) & 7)) return SDN_E_INVALID;
  if (rows == 0) return SDN_OK;
#define SDN_RS_LAUNCH(NQ, R)                                                                                          \
  hipLaunchKernelGGL((k_row_stats<T, NQ, R>), dim3((unsigned)((rows + 4 * (R) - 1) / (4 * (R)))), dim3(THREADS), 0,      \
                     (hipStream_t)stream, (const unsigned short*)x, (long)rows, c, eps, out)
  if (c <= 512) SDN_RS_LAUNCH(1, 4);
  else if (c <= 1024) SDN_RS_LAUNCH(2, 2);
  else SDN_RS_LAUNCH(4, 1);
#undef SDN_RS_LAUNCH
  return sdn_launch_status();
}

}  // namespace sdn_norm_detail

// internal (sdn_ops.h): timestep features with the timestep read from device memory; a one-float store
int sdn_temb_from_device(int dtype, const float* t_dev, int batch, int dim, void* out, void* stream) {
  if (!t_dev) return SDN_E_INVALID;
  return dtype == 1 ? sdn_norm_detail::temb_impl<SdnF16>(0.f, batch, dim, out, stream, t_dev)
                    : sdn_norm_detail::temb_impl<SdnBF16>(0.f, batch, dim, out, stream, t_dev);
}
namespace { __global__ void k_set_scalar(float* dst, float v) { *dst = v; } }
int sdn_set_scalar(float* dst, float v, void* stream) {
  if (!dst) return SDN_E_INVALID;
  hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, (hipStream_t)stream, dst, v);
  return sdn_launch_status();
}
using namespace sdn_norm_detail;

#define SDN_NORM_ENTRY(SUF, T)                                                                                          \
  extern "C" int sdn_groupnorm_##SUF(const void* x, const void* x2, int32_t batch, int32_t hw, int32_t c1, int32_t c2, \
                                     int32_t groups, float eps, int32_t silu, const float* gamma, const float* beta,   \
                                     void* out, float* stats_ws, void* stream) {                                        \
    return groupnorm_impl<T>(x, x2, batch, hw, c1, c2, groups, eps, silu, gamma, beta, out, stats_ws, stream);         \
  }                                                                                                                     \
  extern "C" int sdn_groupnorm_cols_##SUF(const void* x, const void* x2, int32_t batch, int32_t hw, int32_t c1,         \
                                          int32_t c2, int32_t groups, float eps, int32_t silu, const float* gamma,      \
                                          const float* beta, void* out, float* stats_ws, const float* cols1,            \
                                          const float* cols2, void* stream) {                                           \
    if (!cols1) return SDN_E_INVALID;                                                                                   \
    return groupnorm_impl<T>(x, x2, batch, hw, c1, c2, groups, eps, silu, gamma, beta, out, stats_ws, stream, cols1,    \
                             cols2);                                                                                    \
  }                                                                                                                     \
  extern "C" int sdn_layernorm_##SUF(const void* x, int64_t rows, int32_t c, float eps, const float* gamma,            \
                                     const float* beta, void* out, void* stream) {                                      \
    return layernorm_impl<T>(x, rows, c, eps, gamma, beta, out, stream);                                                \
  }                                                                                                                     \
  extern "C" int sdn_layernorm_mod_##SUF(const void* x, int64_t rows, int32_t c, float eps, const float* scale,        \
                                         const float* shift, int32_t ld_mod, int32_t rows_per_batch, void* out,        \
                                         void* stream) {                                                                \
    return layernorm_impl<T>(x, rows, c, eps, scale, shift, out, stream, 1, rows_per_batch, ld_mod);                    \
  }                                                                                                                     \
  extern "C" int sdn_patchify_##SUF(const float* lat, int32_t B, int32_t C, int32_t H, int32_t W, int32_t p, void* out, \
                                    void* stream) {                                                                     \
    return patchify_impl<T>(lat, B, C, H, W, p, out, stream);                                                           \
  }                                                                                                                     \
  extern "C" int sdn_conv_in_##SUF(const float* lat, const void* w, const float* bias, int32_t batch, int32_t cin,     \
                                   int32_t h, int32_t wd, int32_t cout, void* out, void* stream) {                      \
    return conv_in_impl<T>(lat, w, bias, batch, cin, h, wd, cout, out, stream);                                         \
  }                                                                                                                     \
  extern "C" int sdn_timestep_embed_##SUF(float timestep, int32_t batch, int32_t dim, void* out, void* stream) {       \
    return temb_impl<T>(timestep, batch, dim, out, stream);                                                             \
  }
SDN_NORM_ENTRY(bf16, SdnBF16)
SDN_NORM_ENTRY(f16, SdnF16)

extern "C" int sdn_row_stats_bf16(const void* x, int64_t rows, int32_t c, float eps, float* out, void* stream) {
  return sdn_norm_detail::row_stats_impl<SdnBF16>(x, rows, c, eps, out, stream);
}
extern "C" int sdn_row_stats_f16(const void* x, int64_t rows, int32_t c, float eps, float* out, void* stream) {
  return sdn_norm_detail::row_stats_impl<SdnF16>(x, rows, c, eps, out, stream);
}

extern "C" int sdn_unpatchify_f32(const float* tok, int32_t B, int32_t C, int32_t H, int32_t W, int32_t p, float* out,
                                  void* stream) {
  if (!tok || !out || B < 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0 || H % p || W % p) return SDN_E_INVALID;
  if (B == 0) return SDN_OK;
  const long total = (long)B * C * H * W;
  long grid = (total + THREADS - 1) / THREADS;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_unpatchify, dim3((unsigned)grid), dim3(THREADS), 0, (hipStream_t)stream, tok, B, C, H, W, p, out);
  return sdn_launch_status();
}
