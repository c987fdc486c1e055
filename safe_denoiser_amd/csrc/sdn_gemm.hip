// bf16 MFMA GEMM with fused im2col (3x3 conv as implicit GEMM) and fused epilogues -- the kernel ~85 % of the
// SD-v1.4 UNet FLOPs run through (rows U3-U6: convs, attention projections, GEGLU feed-forward, 1x1 convs).
//
//   C[M,N] = A[M,K] . W[N,K]^T  (+ bias[n] + rowbias[b(m),n] + residual[m,n]) -> act -> bf16 | f32 | f32 NCHW
//
// Layout (MI355X-first): activations are NHWC bf16, so a [B,H,W,C] feature map IS the [B*H*W, C] token matrix
// the transformer blocks use -- no NCHW<->sequence transposes anywhere.  Weights are [N][K] bf16 with K
// contiguous; conv weights are pre-permuted to [Cout][ky][kx][Cin] so one BK=64 k-tile lies inside one tap.
//
// Tile: BM=128 x BN=32*NREP (NREP=5 -> 160 divides every SD-v1.4 channel count 320/640/1280 exactly) x BK=64,
// 4 waves as 2(M) x 2(N); each wave owns 64 x 16*NREP via v_mfma_f32_16x16x32_bf16 in the SWAPPED orientation
// (weights = A operand, activations = B operand) so a lane ends up with 4 consecutive output channels of one
// row -> 8-byte packed stores and float4 bias loads.  LDS tiles are [row][8 x 16 B] with chunk ^ (row & 7)
// swizzle: conflict-free ds_read_b128 fragment reads (checked by brute force over the gfx950 lane groups).
// Pipeline: register-staged double buffer (global loads for tile k+1 issued before the MFMAs of tile k, LDS
// write after them; one barrier per k-tile).  Zero padding of the conv halo and of the M tail is done by
// predicated loads (zeros), so no padded copies of the activations exist in HBM.
#include "sdn_common.h"
#include "sdn_ops.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int BM = 128, BK = 64, THREADS = 256;

struct GemmArgs {
  const __bf16* a;  const __bf16* a2;  const __bf16* w;
  const float* bias;  const float* rowbias;  const __bf16* residual;  void* out;
  int M, N, K, K1;
  int a_mode, Hs, Ws, Cin, Ho, Wo, stride, upsample;
  int act, out_kind, rows_per_batch, ld_rowbias, n_valid, ldc;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 p = {(__bf16)lo, (__bf16)hi};                 // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
  return *reinterpret_cast<unsigned*>(&p);
}
__device__ __forceinline__ float silu_f(float v) { return v / (1.f + __expf(-v)); }
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }

// byte offset of 16-B chunk c of row r inside a [rows][8 chunks] tile
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

template <int NREP>
__global__ void __launch_bounds__(THREADS, 2)
k_gemm(const GemmArgs g) {
  constexpr int BN = 32 * NREP;
  constexpr int A_CHUNKS = BM * 8 / THREADS;                 // 4
  constexpr int W_CHUNKS = (BN * 8 + THREADS - 1) / THREADS; // 5 @160, 4 @128, 1 @32
  constexpr int STAGE = (BM + BN) * 128;                     // bytes per pipeline stage
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  // ---- XCD-aware tile mapping: each XCD (blocks b, b+8, ...) gets a contiguous run of tiles; inside the run
  //      the n-tiles of one m-tile are adjacent, so the A panel is fetched once per XCD L2 ----
  const int nt = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int bid = blockIdx.x, q = nt >> 3, r = nt & 7, x = bid & 7;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  // ---- per-thread staging coordinates (fixed across the k loop) ----
  const int chunk = tid & 7;                 // 16-B chunk inside the 128-B k-row
  const int row0 = tid >> 3;                 // 0..31 ; rows row0 + 32*i
  // A rows: decode (b, oy, ox) once
  int a_valid[A_CHUNKS];
  int a_b[A_CHUNKS], a_y[A_CHUNKS], a_x[A_CHUNKS];
  long a_rowoff[A_CHUNKS];
#pragma unroll
  for (int i = 0; i < A_CHUNKS; ++i) {
    const int m = m0 + row0 + 32 * i;
    a_valid[i] = m < g.M;
    const int mm = a_valid[i] ? m : 0;
    if (g.a_mode == 1) {
      const int hw = g.Ho * g.Wo;
      const int b = mm / hw, p = mm - b * hw;
      a_b[i] = b; a_y[i] = p / g.Wo; a_x[i] = p - a_y[i] * g.Wo;
      a_rowoff[i] = 0;
    } else {
      a_b[i] = a_y[i] = a_x[i] = 0;
      a_rowoff[i] = (long)mm;
    }
  }

  u32x4 ra[A_CHUNKS], rw[W_CHUNKS];

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    if (g.a_mode == 1) {
      const int tap = k0 / g.Cin, c0 = k0 - tap * g.Cin;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const int Hi = g.upsample ? g.Hs * 2 : g.Hs, Wi = g.upsample ? g.Ws * 2 : g.Ws;
#pragma unroll
      for (int i = 0; i < A_CHUNKS; ++i) {
        int iy = a_y[i] * g.stride + dy, ix = a_x[i] * g.stride + dx;
        const bool ok = a_valid[i] && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
        if (g.upsample) { iy >>= 1; ix >>= 1; }
        if (ok) {
          const long off = (((long)a_b[i] * g.Hs + iy) * g.Ws + ix) * g.Cin + c0 + chunk * 8;
          ra[i] = *reinterpret_cast<const u32x4*>(g.a + off);
        } else {
          ra[i] = (u32x4){0u, 0u, 0u, 0u};
        }
      }
    } else {
      const __bf16* base = g.a; int ld = g.K1, kk = k0;
      if (k0 >= g.K1) { base = g.a2; ld = g.K - g.K1; kk = k0 - g.K1; }
#pragma unroll
      for (int i = 0; i < A_CHUNKS; ++i) {
        if (a_valid[i]) ra[i] = *reinterpret_cast<const u32x4*>(base + a_rowoff[i] * ld + kk + chunk * 8);
        else ra[i] = (u32x4){0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int i = 0; i < W_CHUNKS; ++i) {
      const int r = row0 + 32 * i;
      if (r < BN) rw[i] = *reinterpret_cast<const u32x4*>(g.w + (long)(n0 + r) * g.K + k0 + chunk * 8);
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* sa = smem + buf * STAGE;
    unsigned char* sw = sa + BM * 128;
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) *reinterpret_cast<u32x4*>(sa + lds_off(row0 + 32 * i, chunk)) = ra[i];
#pragma unroll
    for (int i = 0; i < W_CHUNKS; ++i) {
      const int r = row0 + 32 * i;
      if (r < BN) *reinterpret_cast<u32x4*>(sw + lds_off(r, chunk)) = rw[i];
    }
  };

  f32x4 acc[4][NREP];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NREP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const unsigned char* sa = smem + buf * STAGE + (wm * 64) * 128;
    const unsigned char* sw = smem + buf * STAGE + BM * 128 + (wn * 16 * NREP) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fw[NREP];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sa + lds_off(i * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < NREP; ++j) fw[j] = *reinterpret_cast<const bf16x8*>(sw + lds_off(j * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds, for output row m = ..+fr, channels n = ..+fq*4 + {0,1,2,3} ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= g.M) continue;
    const int b = g.rows_per_batch > 0 ? m / g.rows_per_batch : 0;
    if (g.act == 2) {                                           // GEGLU: even block = value, odd block = gate
#pragma unroll
      for (int j = 0; j + 1 < NREP; j += 2) {
        const int np = n0 + wn * 16 * NREP + j * 16 + fq * 4;   // packed (interleaved) column of the value half
        f32x4 hv = acc[i][j], gv = acc[i][j + 1];
        if (g.bias) {
          const float4 bh = *reinterpret_cast<const float4*>(g.bias + np);
          const float4 bg = *reinterpret_cast<const float4*>(g.bias + np + 16);
          hv[0] += bh.x; hv[1] += bh.y; hv[2] += bh.z; hv[3] += bh.w;
          gv[0] += bg.x; gv[1] += bg.y; gv[2] += bg.z; gv[3] += bg.w;
        }
        const int no = ((n0 + wn * 16 * NREP + j * 16) >> 1) + fq * 4;
        uint2 pk;
        pk.x = pack_bf16(hv[0] * gelu_erf(gv[0]), hv[1] * gelu_erf(gv[1]));
        pk.y = pack_bf16(hv[2] * gelu_erf(gv[2]), hv[3] * gelu_erf(gv[3]));
        *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(g.out) + (long)m * g.ldc + no) = pk;
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      const int n = n0 + wn * 16 * NREP + j * 16 + fq * 4;
      f32x4 v = acc[i][j];
      if (g.bias) {
        const float4 bb = *reinterpret_cast<const float4*>(g.bias + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
      }
      if (g.rowbias) {
        const float4 rb = *reinterpret_cast<const float4*>(g.rowbias + (long)b * g.ld_rowbias + n);
        v[0] += rb.x; v[1] += rb.y; v[2] += rb.z; v[3] += rb.w;
      }
      if (g.residual) {
        const uint2 rr = *reinterpret_cast<const uint2*>(g.residual + (long)m * g.ldc + n);
        v[0] += bf2f(rr.x & 0xffff); v[1] += bf2f(rr.x >> 16); v[2] += bf2f(rr.y & 0xffff); v[3] += bf2f(rr.y >> 16);
      }
      if (g.act == 1) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
      if (g.out_kind == 0) {
        if (n + 3 < g.n_valid) {
          uint2 pk; pk.x = pack_bf16(v[0], v[1]); pk.y = pack_bf16(v[2], v[3]);
          *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(g.out) + (long)m * g.ldc + n) = pk;
        } else {
          for (int e = 0; e < 4; ++e)
            if (n + e < g.n_valid) reinterpret_cast<__bf16*>(g.out)[(long)m * g.ldc + n + e] = (__bf16)v[e];
        }
      } else if (g.out_kind == 1) {
        for (int e = 0; e < 4; ++e)
          if (n + e < g.n_valid) reinterpret_cast<float*>(g.out)[(long)m * g.ldc + n + e] = v[e];
      } else {                                                  // f32 NCHW: out[b][n][p]
        const int p = m - b * g.rows_per_batch;
        for (int e = 0; e < 4; ++e)
          if (n + e < g.n_valid)
            reinterpret_cast<float*>(g.out)[((long)b * g.n_valid + n + e) * g.rows_per_batch + p] = v[e];
      }
    }
  }
}

template <int NREP>
int launch(const GemmArgs& ga, hipStream_t st) {
  const int grid = ga.tiles_m * ga.tiles_n;
  hipLaunchKernelGGL((k_gemm<NREP>), dim3(grid), dim3(THREADS), 0, st, ga);
  return sdn_launch_status();
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Picks the widest N tile that divides N (160 for SD-v1.4 widths, 128 for GEGLU / MMDiT widths, 64, 32).
int sdn_gemm_pick_nrep(int n_padded, int act) {
  if (act == SDN_ACT_GEGLU) return (n_padded % 128 == 0) ? 4 : ((n_padded % 64 == 0) ? 2 : 0);
  if (n_padded % 160 == 0) return 5;
  if (n_padded % 128 == 0) return 4;
  if (n_padded % 64 == 0) return 2;
  if (n_padded % 32 == 0) return 1;
  return 0;
}

extern "C" int sdn_gemm_bf16(const sdn_gemm_desc* d, const void* a, const void* a2, const void* w,
                             const float* bias, const float* rowbias, const void* residual, void* out,
                             void* stream) {
  if (!d || !a || !w || !out) return SDN_E_INVALID;
  if (d->M < 0 || d->N <= 0 || d->K <= 0 || (d->K % BK) != 0) return SDN_E_INVALID;
  if (d->M == 0) return SDN_OK;
  const int n_valid = d->n_valid > 0 ? d->n_valid : d->N;
  if (n_valid > d->N) return SDN_E_INVALID;
  const int nrep = sdn_gemm_pick_nrep(d->N, d->act);
  if (nrep == 0) return SDN_E_INVALID;
  if (!al16(a) || !al16(w) || (a2 && !al16(a2)) || (residual && (reinterpret_cast<uintptr_t>(residual) & 7)) ||
      (reinterpret_cast<uintptr_t>(out) & 7) || (bias && !al16(bias)) || (rowbias && !al16(rowbias)))
    return SDN_E_INVALID;
  GemmArgs g{};
  g.a = (const __bf16*)a; g.a2 = (const __bf16*)a2; g.w = (const __bf16*)w;
  g.bias = bias; g.rowbias = rowbias; g.residual = (const __bf16*)residual; g.out = out;
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.a_mode = d->a_mode;
  if (d->a_mode == SDN_A_PLAIN) {
    g.K1 = (d->K1 > 0 && d->K1 < d->K) ? d->K1 : d->K;
    if (g.K1 != d->K && (!a2 || (g.K1 % BK) != 0)) return SDN_E_INVALID;
  } else if (d->a_mode == SDN_A_CONV3X3) {
    if (d->Cin <= 0 || (d->Cin % BK) != 0 || d->K != 9 * d->Cin || d->Hs <= 0 || d->Ws <= 0 || d->Ho <= 0 ||
        d->Wo <= 0 || (d->stride != 1 && d->stride != 2) || d->M % (d->Ho * d->Wo) != 0)
      return SDN_E_INVALID;
    const int Hi = d->upsample ? 2 * d->Hs : d->Hs, Wi = d->upsample ? 2 * d->Ws : d->Ws;
    if ((Hi + 2 - 3) / d->stride + 1 != d->Ho || (Wi + 2 - 3) / d->stride + 1 != d->Wo) return SDN_E_INVALID;
    g.K1 = d->K; g.Hs = d->Hs; g.Ws = d->Ws; g.Cin = d->Cin; g.Ho = d->Ho; g.Wo = d->Wo; g.stride = d->stride;
    g.upsample = d->upsample;
  } else {
    return SDN_E_INVALID;
  }
  if (d->act < 0 || d->act > 2 || d->out_kind < 0 || d->out_kind > 2) return SDN_E_INVALID;
  if (d->act == SDN_ACT_GEGLU && (d->out_kind != SDN_OUT_BF16 || rowbias || residual || n_valid != d->N))
    return SDN_E_INVALID;
  if ((rowbias || d->out_kind == SDN_OUT_F32_NCHW) && d->rows_per_batch <= 0) return SDN_E_INVALID;
  g.act = d->act; g.out_kind = d->out_kind; g.rows_per_batch = d->rows_per_batch; g.ld_rowbias = d->ld_rowbias;
  g.n_valid = n_valid;
  g.ldc = d->ldc > 0 ? d->ldc : (d->act == SDN_ACT_GEGLU ? d->N / 2 : n_valid);
  if (d->out_kind == SDN_OUT_BF16 && (g.ldc & 3)) return SDN_E_INVALID;
  const int bn = 32 * nrep;
  g.tiles_m = (d->M + BM - 1) / BM; g.tiles_n = d->N / bn;
  hipStream_t st = (hipStream_t)stream;
  switch (nrep) {
    case 5: return launch<5>(g, st);
    case 4: return launch<4>(g, st);
    case 2: return launch<2>(g, st);
    default: return launch<1>(g, st);
  }
}
