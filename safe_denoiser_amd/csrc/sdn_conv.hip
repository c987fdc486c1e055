// 3x3 stride-1 convolution on the 256 x 320 tile with the A operand served from an LDS SLAB RING instead of nine DMA
// re-fetches (rows U3: the resnet convs of the 64 x 64 ... 16 x 16 levels -- half of the SD-v1.4 forward).
//
// Why: the forward runs the chip at its power limit, and what it responds to is bytes, not issue slots.  In the
// implicit-GEMM form (sdn_gemm.hip) every k-tile DMAs a 256 x 64 A tile = one tap's shifted window of the input map: the same
// pixels cross L2 -> LDS nine times per 64-channel chunk.  A timing-only ablation that fetched A for one tap in nine (wrong
// results, right traffic) made the SUSTAINED forward 5.9 % faster (106.6 -> 100.3 ms) -- more than any kernel-level change of
// round 2.  Here a 256-row tile is R = 256 / W whole image rows (W = 64, 32, 16: H W is a multiple of 256, so a tile never
// straddles images); per 64-channel chunk the (R + 2) x W input window ("slab": rows y0 - 1 ... y0 + R) is DMA'd ONCE and all
// nine taps read their fragments from it at shifted addresses.  L2 -> LDS bytes per k-tile: 40 KB of W + 48 KB / 9 of slab
// instead of 40 + 32.
//
// LDS: [W stage 0 | W stage 1 | slab ring of NR = 2R + 2 rows of W x 128 B]  (W = 64: 80 + 80 KB = all 160 KB).
// Slab row j (0 ... R + 1) of chunk c lives in ring slot (c (R + 2) + j) mod NR, pixel x at byte x * 128, 16-byte chunk k at
// ((k ^ (x & 7)) << 4) -- the swizzle of the GEMM tiles with the pixel as the row, applied to the DMA's per-lane SOURCE
// address.  Refill, all behind the per-k-tile vmcnt(0) + barrier: at tap 0 of chunk c every wave issues its 4 pieces of chunk
// c+1's rows 0 ... R-1 (slots of chunk c-1's rows 2 ... R+1, dead since its last tap) and chunk c's own rows R, R+1 (slots of
// chunk c-1's rows 0, 1; first needed at taps 3 and 6).  Left / right halo columns lie outside the image for full-width rows:
// the edge lanes of the dx != 0 taps get zeros by a select after the read; top / bottom halo rows are zero-filled by the
// DMA's range check.
//
// k order (channel chunk outer, tap inner), MFMA shape, accumulator start and epilogue are those of k_gemm_dma: the output is
// bit-identical to the implicit-GEMM path (tests/test_gpu_ops.py), which remains for strides, upsampling and narrow maps.
#include "sdn_gemm_common.h"

namespace sdn_gemm_detail {

#define SDN_STAMP(IDX) {}

template <typename T, int W>
__global__ void __launch_bounds__(512, 1)
k_conv_slab(const GemmArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BN = 320, NREP = 10, WGM = 4, NWAVES = 8, THREADS = 512;
  constexpr int R = BM / W, NR = 2 * R + 2, ROWB = W * 128, PPR = W / 8;   // image rows per tile, ring rows, bytes / pieces per slab row
  constexpr int WST = BN * 128;                              // one k-tile of weights: 40 KB
  constexpr int OFF_RING = 2 * WST;
  constexpr int LDS_BYTES = OFF_RING + NR * ROWB;
  constexpr int NSTAGE = 2, STAGE = LDS_BYTES / 2;           // (names the shared epilogue sizes its scratch with)
  constexpr unsigned OOB = 0x80000000u;
  constexpr int LNF = 0;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int lrow = lane >> 3, lslot = lane & 7;
  const int fr = lane & 15, fq = lane >> 4;
  const int H = W;                                           // square maps (host-checked)
  const int nt = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int vid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7, x = vid & 7;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (vid >> 3);
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int img = m0 / (H * W), y0 = (m0 - img * (H * W)) / W;
  const int Cin = g.Cin, nchunks = Cin / BK, nk = 9 * nchunks;

  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(g.w, (unsigned)((long)g.N * g.K * 2));
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(g.a, (unsigned)((long)(g.M / (H * W)) * H * W * Cin * 2));

  // ---- weights: 40 pieces per k-tile, 5 per wave (as k_gemm_dma) ----
  unsigned w_off[5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
    w_off[i] = (unsigned)(((long)(n0 + (wid * 5 + i) * 8 + lrow) * g.K + (lslot ^ lrow) * 8) * 2);
  auto issue_w = [&](int kt) {                               // k-tile kt = chunk kt / 9, tap kt % 9
    const int c = kt / 9, tap = kt - c * 9;
    unsigned char* sw = smem + (kt & 1) * WST;
    const unsigned kb = (unsigned)((tap * Cin + c * BK) * 2);
#pragma unroll
    for (int i = 0; i < 5; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sw + (wid * 5 + i) * 1024), 16, w_off[i] + kb, 0, 0, 0);
  };
  // ---- slab rows: piece = 8 pixels x 128 B of one row; lane (lrow = pixel in piece, lslot) fetches logical chunk lslot ^ lrow ----
  auto issue_slab_piece = [&](int c, int j, int px8) {       // row j of chunk c, pixels 8 px8 ... 8 px8 + 7
    int slot = (c * (R + 2)) % NR + j;
    if (slot >= NR) slot -= NR;
    const int gy = y0 + j - 1;
    const int px = px8 * 8 + lrow;
    const unsigned off = (gy >= 0 && gy < H) ? (unsigned)((((long)(img * H + gy) * W + px) * Cin + c * BK + (lslot ^ lrow) * 8) * 2) : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(smem + OFF_RING + slot * ROWB + px8 * 1024), 16, off, 0, 0, 0);
  };
  auto issue_slab_main = [&](int c) {                        // rows 0 ... R-1: R PPR = 32 pieces, 4 per wave
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int p = wid * 4 + q; issue_slab_piece(c, p / PPR, p % PPR); }
  };
  auto issue_slab_tail = [&](int c) {                        // rows R, R+1: 2 PPR pieces over the waves
#pragma unroll
    for (int q = 0; q < (2 * PPR + NWAVES - 1) / NWAVES; ++q) {
      const int p = wid + q * NWAVES;
      if (p < 2 * PPR) issue_slab_piece(c, R + p / PPR, p % PPR);
    }
  };

  issue_slab_main(0);
  issue_slab_tail(0);
  issue_w(0);
  // accumulators start at bias (+ the per-sample row bias), as in k_gemm_dma
  f32x4 acc[4][NREP];
  if (g.bias) {
    f32x4 bv[NREP];
#pragma unroll
    for (int j = 0; j < NREP; ++j) bv[j] = *reinterpret_cast<const f32x4*>(g.bias + n0 + wn * 16 * NREP + j * 16 + fq * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NREP; ++j) acc[i][j] = bv[j];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NREP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (g.rowbias) {
    const float* rbp = g.rowbias + (long)img * g.ld_rowbias + n0 + wn * 16 * NREP + fq * 4;   // one sample per tile
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      const f32x4 rb = *reinterpret_cast<const f32x4*>(rbp + j * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] += rb;
    }
  }
  if (g.res_pre) {                                           // 16-bit residual into the accumulators, as k_gemm_dma (sdn_gemm_desc.res_pre)
    const unsigned short* r16 = reinterpret_cast<const unsigned short*>(g.res_pre);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      const unsigned short* rp = r16 + (long)(m < g.M ? m : 0) * g.ldc + n0 + wn * 16 * NREP + fq * 4;
#pragma unroll
      for (int j = 0; j < NREP; ++j) {
        const uint2 rr = *reinterpret_cast<const uint2*>(rp + j * 16);
        acc[i][j][0] += T::to_f(rr.x & 0xffff); acc[i][j][1] += T::to_f(rr.x >> 16);
        acc[i][j][2] += T::to_f(rr.y & 0xffff); acc[i][j][3] += T::to_f(rr.y >> 16);
      }
    }
  }
  if (g.x3_out && g.residual) {                              // bf16x3 plan: f32 residual [M, ldc], in the shadow of the first k-tile's DMA
    const float* resf = reinterpret_cast<const float*>(g.residual);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      const float* rp = resf + (long)(m < g.M ? m : 0) * g.ldc + n0 + wn * 16 * NREP + fq * 4;
#pragma unroll
      for (int j = 0; j < NREP; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(rp + j * 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // this wave's output pixels: p = wm 64 + i 16 + fr -> tile row r_i (wave-uniform), column x_i (per lane)
  int xi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xi[i] = (wm * 64 + i * 16 + fr) % W;

  int c = 0, tap = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int ty = tap / 3, tx = tap - ty * 3;
    const int dy = ty - 1, dx = tx - 1;
    const int base = (c * (R + 2)) % NR;
    const unsigned char* sw = smem + (kt & 1) * WST + (wn * 16 * NREP) * 128;
    const bool more = kt + 1 < nk;
    // A fragment addresses of this tap (k-step 0; k-step 1 = the same with address bit 6 flipped)
    unsigned fao[4];                                         // byte offsets inside smem (integers: the XOR below must not turn the
    bool edge[4];                                            // LDS pointer into a generic one)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ri = (wm * 64 + i * 16) / W;                 // wave-uniform tile row of this fragment
      int slot = base + ri + dy + 1;
      if (slot >= NR) slot -= NR;
      int px = xi[i] + dx;
      edge[i] = px < 0 || px >= W;
      px = px < 0 ? 0 : (px >= W ? W - 1 : px);
      fao[i] = (unsigned)(OFF_RING + slot * ROWB + px * 128 + ((fq ^ (px & 7)) << 4));
    }
    typename T::v8 fa[2][4], fw[2][2];
    auto read_fa = [&](int ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[ks][i] = *reinterpret_cast<const typename T::v8*>(smem + (fao[i] ^ (unsigned)(ks * 64)));
      }
    };
    auto zero_edges = [&](int ks) {
      if (dx != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // (compile-time pruning: for W = 64 only the first / last fragment of a row can touch an edge, for W = 32 every other)
          if ((W == 64 && i != 0 && i != 3) ) continue;
          u32x4 v = *reinterpret_cast<u32x4*>(&fa[ks][i]);
          if (edge[i]) v = (u32x4){0u, 0u, 0u, 0u};
          fa[ks][i] = *reinterpret_cast<typename T::v8*>(&v);
        }
      }
    };
    // the younger half of the workgroup issues the next k-tile's DMA at the top of the iteration, its SIMD partners of the older
    // half behind their first k-step's MFMAs (see k_gemm_dma; dbg 32: everyone behind the first k-step as in rounds 2-4, 64: reversed)
    const bool dma_top = !(g.dbg & 32) && ((wid >= 4) != ((g.dbg & 64) != 0));
    auto issue_next = [&]() {
      if (more) issue_w(kt + 1);
      if (tap == 0) {
        if (c + 1 < nchunks) issue_slab_main(c + 1);
        if (c >= 1) issue_slab_tail(c);
      }
    };
    if (dma_top) issue_next();
    read_fa(0);
#pragma unroll
    for (int j = 0; j < 2; ++j) fw[0][j] = *reinterpret_cast<const typename T::v8*>(sw + lds_off(j * 16 + fr, fq));
    zero_edges(0);
#pragma unroll
    for (int gi = 0; gi < 10; ++gi) {                        // W fragments in groups of two, next group's reads ahead (as k_gemm_dma)
      const int ks = gi / 5, gq = gi % 5;
      if (gi + 1 < 10) {
        const int ks1 = (gi + 1) / 5, g1 = (gi + 1) % 5;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          fw[(gi + 1) & 1][j] = *reinterpret_cast<const typename T::v8*>(sw + lds_off((2 * g1 + j) * 16 + fr, ks1 * 4 + fq));
      }
      if (gi == 2) { read_fa(1); }
      if (gi == 4) { zero_edges(1); }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][2 * gq + j] = T::mfma16(fw[gi & 1][j], fa[ks][i], acc[i][2 * gq + j]);
      __builtin_amdgcn_s_setprio(0);
      if (gi == 4 && !dma_top) issue_next();                  // DMA behind the first k-step's MFMAs (see k_gemm_dma)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (++tap == 9) { tap = 0; ++c; }
  }

  // ---- epilogue: the shared GEMM epilogue (two staging passes, residual slab, column sums) ----
  [[maybe_unused]] const float ln_mu[4] = {0.f, 0.f, 0.f, 0.f}, ln_rs[4] = {0.f, 0.f, 0.f, 0.f};   // (names of the epilogue's discarded LayerNorm branch)
  constexpr int CW_PAD = (BN + 8) * 2;
  const bool staged = true;
  const int out_cols = BN;
  const bool res_lds = g.res_lds != 0;
  const int CW = res_lds ? BN * 2 : CW_PAD;
  const bool lean = !g.rowgate && g.act == 0 && (res_lds || !g.residual);
  const bool lean_gelu = false, lean_gate = false;
  const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(g.residual ? g.residual : g.a, g.res_bytes);
  constexpr int PASSES = 2, WM_PER_PASS = WGM / PASSES, ROWS_PER_PASS = 64 * WM_PER_PASS;
  static_assert(ROWS_PER_PASS * CW_PAD + 8 * BN * 8 <= LDS_BYTES, "staged pass and the column-sum scratch must fit the LDS");
#define SDN_PASS 0
#include "sdn_gemm_epilogue.inc"
#undef SDN_PASS
#define SDN_PASS 1
#include "sdn_gemm_epilogue.inc"
#undef SDN_PASS
  (void)lean_gelu; (void)lean_gate; (void)staged; (void)NSTAGE; (void)STAGE; (void)lean;
#endif  // __HIP_DEVICE_COMPILE__
}

template <typename T>
static int launch_slab(const GemmArgs& g, hipStream_t st) {
  const int grid = g.tiles_m * g.tiles_n;
  switch (g.Ws) {
    case 64: hipLaunchKernelGGL((k_conv_slab<T, 64>), dim3(grid), dim3(512), 0, st, g); break;
    case 32: hipLaunchKernelGGL((k_conv_slab<T, 32>), dim3(grid), dim3(512), 0, st, g); break;
    case 16: hipLaunchKernelGGL((k_conv_slab<T, 16>), dim3(grid), dim3(512), 0, st, g); break;
    default: return SDN_GEMM_NOT_SLAB;
  }
  return sdn_launch_status();
}

// Launches the slab form when the convolution suits it; SDN_GEMM_NOT_SLAB = the caller keeps the implicit-GEMM kernel.
int dispatch_conv_slab(int dtype, const GemmArgs& g, hipStream_t st) {
  if (g.a_mode != 1 || g.Hs != g.Ws || g.Ho != g.Hs || g.Wo != g.Ws ||
      !sdn_conv_slab_shape_ok(g.M, g.N, g.Cin, g.Ws, g.stride, g.upsample, g.conv_off, g.x3_out ? SDN_OUT_BF16 : g.out_kind, g.n_valid))
    return SDN_GEMM_NOT_SLAB;
  if (g.kt_per_split != 0 || g.rowgate || g.act != 0 || g.stamps || (g.dbg & ~96) || (g.residual && !g.res_lds && !g.x3_out) ||
      (g.rowbias && g.rows_per_batch != g.Hs * g.Ws))
    return SDN_GEMM_NOT_SLAB;
  return dtype == 0 ? launch_slab<SdnBF16>(g, st) : launch_slab<SdnF16>(g, st);
}

}  // namespace sdn_gemm_detail

// Shape part of the slab form's applicability (the plan builder labels its launches with it; sdn_ops.h).
int sdn_conv_slab_shape_ok(int M, int N, int Cin, int side, int stride, int upsample, int asym_pad, int out_kind, int n_valid) {
  if (stride != 1 || upsample || asym_pad || (side != 64 && side != 32 && side != 16)) return 0;
  if (N % 320 != 0 || Cin % 64 != 0 || out_kind != SDN_OUT_BF16 || (n_valid != 0 && n_valid != N) || M % 256 != 0) return 0;
  return (long)(M / 256) * (N / 320) >= 192;
}
