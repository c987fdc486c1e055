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
// Pipeline: two LDS stages filled by LDS-DMA (tile k+1 in flight during the MFMAs of tile k; one barrier per
// k-tile).  Zero padding of the conv halo and of the M tail comes from the buffer descriptor's range check, so
// no padded copies of the activations exist in HBM.
#include <stdlib.h>

#include <type_traits>

#include "sdn_common.h"
#include "sdn_ops.h"

static int g_gemm_variant = 0;     // debug A/B switch, see sdn_debug_set_gemm_variant
static unsigned long long* g_gemm_stamps = nullptr;   // diagnostics buffer (4 x grid), see sdn_debug_set_gemm_stamps

#include "sdn_gemm_common.h"

namespace sdn_gemm_detail {

// WGM = waves along M (2 -> 128-row tile, 4 waves, 2 blocks/CU;  4 -> 256-row tile, 8 waves, 1 block/CU).
// The 256 x 320 tile (WGM=4, NREP=10) halves the L2->LDS bytes per MFMA of the 128 x 160 tile: at 2 x 36.9 KB per
// 1280 MFMA-cycles per CU the small tile needs ~57 B/clk/CU of L2 bandwidth -- ~90 % of what a CU can pull.
#define SDN_STAMP(IDX)                                                                                   \
  if (g.stamps && threadIdx.x == 0) {                                                                    \
    unsigned long long t_;                                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
    g.stamps[(long)blockIdx.x * 8 + (IDX)] = t_;                                                         \
  }

// LNF: out = LayerNorm(A) . W^T + b with the LayerNorm folded in: the caller passes W' = W * gamma (per input channel,
// rounded to 16 bit), c[n] = sum_k W'[n,k] and d[n] = sum_k beta[k] W[n,k] + b[n]; the kernel accumulates each row's sum
// and sum of squares from the A fragments it feeds the MFMAs anyway (v_dot2c), and the epilogue applies
// rstd * (acc - mean * c[n]) + d[n].  The normalised activation is never written to or re-read from HBM.
// LNF = 1: row statistics from the fragments (right for narrow N: every n-tile repeats that VALU work -- at N = 8C it
// costs more than the LayerNorm kernel it replaces); LNF = 2: statistics read from g.ln_stats [M][2] = (mean, rstd),
// written by a read-only pre-pass (sdn_row_stats_*), for wide N.
// (H8 = the experimental operand form of DESIGN 10.12; it lives in its OWN kernel symbol, k_gemm_h8, so that the production
//  instances keep their code and registers -- sharing one body cost the fp16 instances 600 spilled registers)
template <typename T, int NREP, int WGM, int NSTAGE, int LNF, bool H8>
__device__ __forceinline__ void gemm_dma_body(const GemmArgs& g) {
#if defined(__HIP_DEVICE_COMPILE__)
  SDN_STAMP(0)
  constexpr int BM = 64 * WGM, THREADS = 128 * WGM, NWAVES = 2 * WGM;
  constexpr int BN = 32 * NREP;
  constexpr int A_PIECES = BM / 8 / NWAVES;                  // 1-KiB pieces per wave: 4
  constexpr int W_PIECES = (BN / 8 + NWAVES - 1) / NWAVES;   // 5 @160/4w or @320/8w, 4 @128/4w, ...
  constexpr int STAGE = (BM + BN) * 128;
  constexpr unsigned OOB = 0x80000000u;                      // > any tensor size handled here
  // NSTAGE = 2: one k-tile of prefetch, 2 workgroups per CU cover each other's waits.  NSTAGE = 4 (grids of at most one
  // workgroup per CU, e.g. the 8x8-level convs, M = 4096): the lone workgroup owns the LDS, so it keeps 3 k-tiles in
  // flight behind counted vmcnt waits and raw barriers instead.
  constexpr int LPI = A_PIECES + W_PIECES;                   // DMA instructions per wave per k-tile (every wave issues all)
  static_assert(NSTAGE == 2 || (BN / 8) % NWAVES == 0, "counted waits need the same number of loads in every wave");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE];

  const int nt = g.tiles_m * g.tiles_n;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: LDS-DMA bases go to M0, no waterfall
  const int wm = wid >> 1, wn = wid & 1;
  const int lrow = lane >> 3;                                // row inside a piece
  // (A persistent tile loop -- one residency of workgroups walking all tiles -- was measured: +-3 %, i.e. workgroup
  //  dispatch is not what the short-K shapes pay for, and it costs 30 VGPRs; one tile per workgroup is kept.)
  int tile;
  {
    const int vid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7, x = vid & 7;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (vid >> 3);
  }
  int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  if (g.panel) {
    // wide N (GEGLU / qkv projections: 6 ... 32 n-tiles): in row-major order the 32 workgroups an XCD runs at a time span 1-5
    // tile rows and EVERY n-tile, i.e. each wave of tiles streams the whole weight matrix (6.5 ... 26 MB) through a 4 MiB L2.
    // Panel order: blocks of 8 tile rows, inside a block panels of g.panel n-tiles, rows fastest -- a wave of 32 tiles is
    // 8 A row-tiles x 4 W column-tiles (2.6 + 1.6 MB at K = 640), and the next wave re-reads the same 8 A row-tiles from L2.
    const int per_blk = 8 * g.tiles_n;
    const int blk = tile / per_blk, rem = tile - blk * per_blk;
    const int rows = min(8, g.tiles_m - blk * 8);
    const int full = rows * g.panel;
    const int p = rem / full, r2 = rem - p * full;
    const int w = min(g.panel, g.tiles_n - p * g.panel);
    const int tl = r2 / w;
    tm = blk * 8 + tl; tn = p * g.panel + (r2 - tl * w);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int lchunk = (lane & 7) ^ lrow;                      // logical 16-B chunk this lane fetches

  // ---- per-lane source rows (fixed across k).  All per-k-tile address work is strength-reduced to
  //      "lane base + wave-uniform delta" (+ a validity bit test for the conv halo) ----
  unsigned a_base[A_PIECES];        // PLAIN: byte offset of (row, lane chunk) in source 1 (or OOB); CONV: centre pixel
  unsigned a_aux[A_PIECES];         // PLAIN: the same offset in source 2 (or OOB)
                                    // CONV: bits 0-8 = tap t reads inside the (virtual) input map, bit 9/10 = parity
                                    //       of the output row / column (nearest-2x upsample)
  const int ld1 = g.a_mode == 1 ? g.Cin : g.K1, ld2 = g.K - g.K1;
  // (this setup runs before the first DMA can be issued -- in-kernel stamps put it at 2-7 k cycles, 4-11 % of a tile -- so
  //  it is kept short: the pieces of a lane are rows m, m+8, m+16, ...: ONE division pair, then increments; the 9-tap halo
  //  mask is an outer product of 3 row bits and 3 column bits)
  if (g.a_mode == 1) {
    const int hw = g.Ho * g.Wo;
    const int Hi = g.upsample ? g.Hs * 2 : g.Hs, Wi = g.upsample ? g.Ws * 2 : g.Ws;
    const int mb = m0 + wid * A_PIECES * 8 + lrow;
    const int mm0 = mb < g.M ? mb : 0;
    int bb = mm0 / hw;
    int oy, ox;
    { const int p = mm0 - bb * hw; oy = p / g.Wo; ox = p - oy * g.Wo; }
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const int m = mb + i * 8;
      const bool ok = m < g.M;
      if (i > 0) {
        if (g.Wo >= 8) {                                     // rows advance by 8: at most one wrap per level
          ox += 8;
          if (ox >= g.Wo) { ox -= g.Wo; if (++oy >= g.Ho) { oy = 0; ++bb; } }
        } else {
          const int mm = ok ? m : 0;
          bb = mm / hw;
          const int p = mm - bb * hw;
          oy = p / g.Wo; ox = p - oy * g.Wo;
        }
      }
      const int cy = oy * g.stride + g.conv_off, cx = ox * g.stride + g.conv_off;   // centre in the virtual (upsampled) input
                                                                 // (conv_off = 1: padding (0,1,0,1), taps start at 2*o)
      const unsigned rb = (cy >= 1 ? 1u : 0u) | (cy < Hi ? 2u : 0u) | (cy + 1 < Hi ? 4u : 0u);
      const unsigned cb = (cx >= 1 ? 1u : 0u) | (cx < Wi ? 2u : 0u) | (cx + 1 < Wi ? 4u : 0u);
      unsigned mask = ((rb & 1u) ? cb : 0u) | ((rb & 2u) ? cb << 3 : 0u) | ((rb & 4u) ? cb << 6 : 0u);
      if (!ok) mask = 0u;
      a_aux[i] = mask | ((unsigned)(cy & 1) << 9) | ((unsigned)(cx & 1) << 10);
      const int sy = g.upsample ? cy >> 1 : cy, sx = g.upsample ? cx >> 1 : cx;
      a_base[i] = ok ? (unsigned)((((bb * g.Hs + sy) * g.Ws + sx) * g.Cin + lchunk * 8) * 2) : 0u;
    }
  } else {
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const int m = m0 + (wid * A_PIECES + i) * 8 + lrow;
      const bool ok = m < g.M;
      a_base[i] = ok ? (unsigned)(((long)m * ld1 + lchunk * 8) * 2) : OOB;
      a_aux[i] = (g.a2 && ok) ? (unsigned)(((long)m * ld2 + lchunk * 8) * 2) : OOB;
    }
  }
  unsigned w_off[W_PIECES];
  bool w_ok[W_PIECES];
#pragma unroll
  for (int i = 0; i < W_PIECES; ++i) {
    const int piece = wid * W_PIECES + i;
    w_ok[i] = piece * 8 < BN;
    w_off[i] = (unsigned)(((long)(n0 + piece * 8 + lrow) * g.K + lchunk * 8) * 2);
  }
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(g.w, (unsigned)((long)g.N * g.K * 2));
  const long a_rows = g.a_mode == 1 ? (long)(g.M / (g.Ho * g.Wo)) * g.Hs * g.Ws : (long)g.M;
  const __amdgpu_buffer_rsrc_t rs_a1 = make_rsrc(g.a, (unsigned)(a_rows * ld1 * 2));
  const __amdgpu_buffer_rsrc_t rs_a2 = make_rsrc(g.a2 ? g.a2 : g.a, (unsigned)(g.a2 ? a_rows * ld2 * 2 : 0));

  // k-tile cursor (issue() is always called for consecutive k-tiles)
  // split-K: this workgroup's slice of the k loop starts at k-tile kt0 (k order of a conv = channel chunk outer, tap inner)
  const int kt0 = g.kt_per_split > 0 ? (int)blockIdx.y * g.kt_per_split : 0;
  int cur_k0 = kt0 * BK, cur_c0 = (kt0 / 9) * BK, cur_ty = (kt0 % 9) / 3, cur_tx = kt0 % 3, w_k0 = 0;
  auto issue_aw = [&](int buf, bool with_w) {
    unsigned char* sa = smem + buf * STAGE;
    unsigned char* sw = sa + BM * 128;
    if (g.a_mode == 1) {
      const int dy = cur_ty - 1, dx = cur_tx - 1, tap = cur_ty * 3 + cur_tx;
      if (!g.upsample) {
        const int delta = ((dy * g.Ws + dx) * g.Cin + cur_c0) * 2;                    // wave-uniform
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
          const unsigned off = ((a_aux[i] >> tap) & 1u) ? a_base[i] + (unsigned)delta : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, (lds_ptr_t)(sa + (wid * A_PIECES + i) * 1024), 16, off, 0, 0, 0);
        }
      } else {                                      // nearest-2x: stored row = (y + dy) >> 1 = (y >> 1) + ((y & 1) + dy) >> 1
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
          const int sy = ((int)((a_aux[i] >> 9) & 1u) + dy) >> 1, sx = ((int)((a_aux[i] >> 10) & 1u) + dx) >> 1;
          const int delta = ((sy * g.Ws + sx) * g.Cin + cur_c0) * 2;
          const unsigned off = ((a_aux[i] >> tap) & 1u) ? a_base[i] + (unsigned)delta : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, (lds_ptr_t)(sa + (wid * A_PIECES + i) * 1024), 16, off, 0, 0, 0);
        }
      }
      // k order = channel-chunk OUTER, tap INNER: the 9 shifted re-reads of one 64-channel slice of the input rows
      // are consecutive k-tiles, so the slice (49 KB per tile at 64x64) is still in the XCD's L2 when it is re-read;
      // tap-outer order kept the whole 320..2560-channel halo live (8 MB per XCD > 4 MiB L2) and re-fetched it 9x.
      w_k0 = (cur_ty * 3 + cur_tx) * g.Cin + cur_c0;
      if (++cur_tx == 3) { cur_tx = 0; if (++cur_ty == 3) { cur_ty = 0; cur_c0 += BK; } }
    } else {
      if (cur_k0 >= g.K1) {
        const unsigned kb = (unsigned)((cur_k0 - g.K1) * 2);
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (lds_ptr_t)(sa + (wid * A_PIECES + i) * 1024), 16, a_aux[i] + kb, 0, 0, 0);
      } else {
        const unsigned kb = (unsigned)(cur_k0 * 2);
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, (lds_ptr_t)(sa + (wid * A_PIECES + i) * 1024), 16, a_base[i] + kb, 0, 0, 0);
      }
    }
    if (!with_w) return;
#pragma unroll
    for (int i = 0; i < W_PIECES; ++i) {
      if (w_ok[i])
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sw + (wid * W_PIECES + i) * 1024), 16,
                                                 w_off[i] + (unsigned)((g.a_mode == 1 ? w_k0 : cur_k0) * 2), 0, 0, 0);
    }
    cur_k0 += BK;
  };
  auto issue = [&](int buf) { issue_aw(buf, true); };
  auto issue_w_only = [&](int buf) {                        // second half of a split issue (issue_aw(buf, false) came first)
    unsigned char* sw = smem + buf * STAGE + BM * 128;
#pragma unroll
    for (int i = 0; i < W_PIECES; ++i) {
      if (w_ok[i])
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sw + (wid * W_PIECES + i) * 1024), 16,
                                                 w_off[i] + (unsigned)((g.a_mode == 1 ? w_k0 : cur_k0) * 2), 0, 0, 0);
    }
    cur_k0 += BK;
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int nk = g.kt_per_split > 0 ? min(g.kt_per_split, g.K / BK - kt0) : g.K / BK;
  SDN_STAMP(7)
  // 2-stage form: the first k-tile's DMA goes out BEFORE the bias loads, which then ride in its shadow (the vmcnt(0)
  // below covers both).  The 4-stage form waits with counted vmcnt, so its bias loads stay ahead of the DMA queue.
  if constexpr (NSTAGE == 2) issue(0);
  // Accumulators start at bias (+ the per-sample row bias): the loads overlap the first k-tile's DMA instead of sitting,
  // one L2 round trip per fragment, in the epilogue (in-kernel stamps: the epilogue was 26-61 % of a workgroup's life).
  // acc[i][j][e] <-> row m0 + wm*64 + i*16 + fr, column n0 + wn*16*NREP + j*16 + fq*4 + e.
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      const int b = m < g.M ? m / g.rows_per_batch : 0;
      const float* rbp = g.rowbias + (long)b * g.ld_rowbias + n0 + wn * 16 * NREP + fq * 4;
#pragma unroll
      for (int j = 0; j < NREP; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(rbp + j * 16);
    }
  }

  if (g.res_pre) {                                           // 16-bit residual, same place (8 bytes per lane and fragment)
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

  if constexpr (NSTAGE == 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else {                                                   // host guarantees nk >= NSTAGE
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue(s);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LPI) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  SDN_STAMP(1)

  [[maybe_unused]] float ln_s1[4] = {0.f, 0.f, 0.f, 0.f}, ln_s2[4] = {0.f, 0.f, 0.f, 0.f};   // LNF 1: per-row sum / sum of squares
  [[maybe_unused]] float ln_mu[4], ln_rs[4];
  if constexpr (LNF == 2) {                                  // pre-pass statistics: fetched under the k loop
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      const float2 st2 = *reinterpret_cast<const float2*>(g.ln_stats + 2 * (long)(m < g.M ? m : 0));
      ln_mu[i] = st2.x; ln_rs[i] = st2.y;
    }
  }
  auto k_iter = [&](const int kt, auto f8_c) __attribute__((always_inline)) {
    const int buf = kt & (NSTAGE - 1);
    // WHERE the next k-tile's DMA is issued.  At the top of the iteration both waves of a SIMD come out of the barrier and spend
    // their first ~1 k cycles issuing 9 DMA pieces each (100-185 cycles apiece next to fragment reads; the conv form adds a
    // halo select per piece) with the matrix pipe idle.  On the 256-row tile the pieces go out BEHIND MFMAs already in the
    // pipe instead: plain A: the A pieces after the first half of the first k-step's MFMAs and the W pieces after its second
    // half (+3...9 % on the K = 640...2560 shapes over the top-of-iteration placement, +2...4 % more from the split); 3x3 conv:
    // everything after the whole first k-step (+7...9 %; the split is neutral there).  Later positions lose.  The 128-row tile
    // (two workgroups per CU cover each other) loses 2-7 % with any late position.  (tools/bench_gemm.py VARIANTS=0,64,128.)
    int ipos = (NSTAGE == 2 && WGM == 4 && !(g.dbg & 8)) ? ((g.dbg & 4) ? 1 : (g.a_mode == 1 ? 2 : ((g.act == 2 && g.K >= 1280) ? 1 : 4))) : 0;   // dbg 8: top of the iteration, 4: unsplit (A/B)
    // Round 5 -- the two halves of the workgroup issue at DIFFERENT points.  The two waves of a SIMD (wid, wid + 4) used to run the
    // same phase order, so their DMA-issue stretches (9 pieces x 100-185 cycles = about as long as a wave's own 80 MFMAs of a
    // k-tile) coincided and the matrix pipe idled under both.  Now the YOUNGER half (waves 4-7: the arbitration losers of every
    // MFMA burst, MI355X_MICROARCH "Two waves per SIMD" item 4) issues the next k-tile's pieces at the TOP of the iteration, while
    // its SIMD partners of the older half run their first k-step's MFMAs, and the older half issues behind that k-step, while the
    // younger half computes.  Same arithmetic, same bits.  Measured (tools/bench_gemm.py VARIANTS=0,512,1024 at B = 192, then
    // tools/ab_sustained.py): 3x3 convs +2.5...4 %, 8x8-level convs +4 %, GEGLU / FF2 / qkv +1...5 %; sustained forward at B = 192
    // 158.5 -> 155.0 ms (-2.2 %); the reverse assignment (older half at the top) gives -0.8 %.  dbg 32: off (A/B), 64: reversed.
    // (dbg 128: the late half keeps the per-shape placement of rounds 2-4 -- split A / W for plain shapes -- instead of "behind the first k-step")
    if (NSTAGE == 2 && WGM == 4 && !(g.dbg & (32 | 8 | 4))) ipos = ((wid >= 4) != ((g.dbg & 64) != 0)) ? 0 : ((g.dbg & 128) ? ipos : 2);
    // (the long-K GEGLU projection of the 16 x 16 level is the one plain shape that prefers the unsplit form: 802 vs 852 us)
    const bool more = kt + NSTAGE - 1 < nk && !(g.dbg & 2);
    if (ipos == 0 && more) issue((kt + NSTAGE - 1) & (NSTAGE - 1));
    const unsigned char* sa = smem + buf * STAGE + (wm * 64) * 128;
    const unsigned char* sw = smem + buf * STAGE + BM * 128 + (wn * 16 * NREP) * 128;
#ifdef SDN_NO_FRAG_PIPE
    constexpr bool kFragPipe = false;                          // A/B build: the compiler's own fragment schedule
#else
    constexpr bool kFragPipe = WGM == 4 && (NREP == 10 || NREP == 8) && LNF != 1;
#endif
    // Experimental h8 operand form: k-tiles past g.h8_t16 hold e4m3 bytes -- 128 per row, the byte geometry of a
    // 16-bit k-tile, so DMA, LDS image and fragment reads are the 16-bit ones -- and one scaled fp8 MFMA (K = 128) per fragment
    // pair replaces the two 16-bit MFMAs: the fragment of k-step 0 and of k-step 1 of a row are the lane's 32 operand bytes
    // (any split of K over the lanes is a valid contraction as long as both operands agree).  Segment 1 = 2^11 lo(a) x q(w): the
    // activation operand carries the block scale 2^-11; segment 2 = q(a) x 2^11 lo(w): the weight operand does.
    constexpr bool f8_tile = decltype(f8_c)::value;          // (H8 only; the two kinds of k-tile run in two loops: see below)
    if constexpr (H8 && f8_tile) {
      {
        typedef __attribute__((ext_vector_type(8))) int i32x8;
        union F8 { typename T::v8 h[2]; i32x8 v; };
        const bool seg1 = kt < g.h8_t16 + (nk - g.h8_t16) / 2;
        const int s_act = seg1 ? 0x74747474 : 0x7f7f7f7f, s_w = seg1 ? 0x7f7f7f7f : 0x74747474;     // e8m0: 0x74 = 2^-11, 0x7f = 1
        // fragment reads two W fragments ahead of the MFMAs that use them; the scheduler is fenced (left alone it requested six W
        // fragments at once and spilled accumulators and the DMA address registers: a scratch reload + vmcnt(0) before every piece)
        F8 av[4], wv[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          av[i].h[0] = *reinterpret_cast<const typename T::v8*>(sa + lds_off(i * 16 + fr, fq));
          av[i].h[1] = *reinterpret_cast<const typename T::v8*>(sa + lds_off(i * 16 + fr, 4 + fq));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          wv[j].h[0] = *reinterpret_cast<const typename T::v8*>(sw + lds_off(j * 16 + fr, fq));
          wv[j].h[1] = *reinterpret_cast<const typename T::v8*>(sw + lds_off(j * 16 + fr, 4 + fq));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NREP; ++j) {
          if (j + 2 < NREP) {
            wv[(j + 2) % 3].h[0] = *reinterpret_cast<const typename T::v8*>(sw + lds_off((j + 2) * 16 + fr, fq));
            wv[(j + 2) % 3].h[1] = *reinterpret_cast<const typename T::v8*>(sw + lds_off((j + 2) * 16 + fr, 4 + fq));
          }
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv[j % 3].v, av[i].v, acc[i][j], 0, 0, 0, s_w, 0, s_act);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          if (j == NREP / 2 - 1 && more && ipos != 0) issue((kt + 1) & 1);
        }
      }
    }
    if constexpr (kFragPipe) {
      if constexpr (!f8_tile) {
      // 256 x 320 tile: the W fragments of a k-step are consumed in 5 groups of 2 (8 MFMAs = 128 matrix-pipe cycles each),
      // and group g+1's two ds_reads are ISSUED BEFORE group g's MFMAs (two alternating 2-fragment buffers; the second
      // k-step's A fragments ride along early), so no MFMA group starts behind a fresh LDS round trip.  Left to the
      // compiler the schedule was "5 reads, wait, 20 MFMAs" four times per k-tile with the pipe draining in each wait.
      typename T::v8 fa[2][4], fw[2][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[0][i] = *reinterpret_cast<const typename T::v8*>(sa + lds_off(i * 16 + fr, fq));
#pragma unroll
      for (int j = 0; j < 2; ++j) fw[0][j] = *reinterpret_cast<const typename T::v8*>(sw + lds_off(j * 16 + fr, fq));
      constexpr int G = NREP / 2;                                // fragment groups per k-step (5 at 320 columns, 4 at 256)
#pragma unroll
      for (int gi = 0; gi < 2 * G; ++gi) {
        const int ks = gi / G, g = gi % G;
        if (gi + 1 < 2 * G) {
          const int ks1 = (gi + 1) / G, g1 = (gi + 1) % G;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            fw[(gi + 1) & 1][j] = *reinterpret_cast<const typename T::v8*>(sw + lds_off((2 * g1 + j) * 16 + fr, ks1 * 4 + fq));
        }
        if (gi == 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[1][i] = *reinterpret_cast<const typename T::v8*>(sa + lds_off(i * 16 + fr, 4 + fq));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][2 * g + j] = T::mfma16(fw[gi & 1][j], fa[ks][i], acc[i][2 * g + j]);
        __builtin_amdgcn_s_setprio(0);
        if (gi == 2 && more) { if (ipos == 1) issue((kt + 1) & 1); else if (ipos == 4) issue_aw((kt + 1) & 1, false); }
        if (gi == G - 1 && more) { if (ipos == 2) issue((kt + 1) & 1); else if (ipos == 4) issue_w_only((kt + 1) & 1); }
      }
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      constexpr int JC = NREP > 5 ? NREP / 2 : NREP;         // W fragments live at once (register budget of the big tile)
      typename T::v8 fa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const typename T::v8*>(sa + lds_off(i * 16 + fr, ks * 4 + fq));
      if constexpr (LNF == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ln_s1[i] = T::dot_ones(fa[i], ln_s1[i]); ln_s2[i] = T::dot_self(fa[i], ln_s2[i]); }
      }
      // (macro, not a lambda: the column-block offset must be a compile-time constant or acc[][] goes to scratch)
#define SDN_MMA_PART(J0)                                                                                              \
      {                                                                                                                \
        typename T::v8 fw[JC];                                                                                         \
        _Pragma("unroll") for (int j = 0; j < JC; ++j)                                                                 \
          fw[j] = *reinterpret_cast<const typename T::v8*>(sw + lds_off(((J0) + j) * 16 + fr, ks * 4 + fq));          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                  \
          _Pragma("unroll") for (int j = 0; j < JC; ++j)                                                               \
            acc[i][(J0) + j] = T::mfma16(fw[j], fa[i], acc[i][(J0) + j]);                                              \
      }
      __builtin_amdgcn_s_setprio(1);                        // the wave that has its fragments issues MFMAs ahead of its SIMD
      SDN_MMA_PART(0)                                        // partner's DMA / fragment-read stream (+0.3 ... 1.1 % on every shape)
      __builtin_amdgcn_s_setprio(0);
      if (ipos == 1 && ks == 0 && more) issue((kt + 1) & 1);
      if (ipos == 4 && ks == 0 && more) issue_aw((kt + 1) & 1, false);     // split: A behind the first MFMA half ...
      __builtin_amdgcn_s_setprio(1);
      if constexpr (NREP > JC) SDN_MMA_PART(JC)
      __builtin_amdgcn_s_setprio(0);
#undef SDN_MMA_PART
      if (ipos == 2 && ks == 0 && more) issue((kt + 1) & 1);
      if (ipos == 4 && ks == 0 && more) issue_w_only((kt + 1) & 1);        // ... W behind the second
    }
    }
    if constexpr (NSTAGE == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      // tile kt+1 must have landed; the tiles issued after it may stay in flight
      const int later = nk - kt - 2;                         // k-tiles issued after tile kt+1 (capped by the ring depth)
      if (later >= NSTAGE - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LPI) : "memory");
      else if (later == 1 && NSTAGE > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this tile's fragment reads are done before its buffer is refilled
      __builtin_amdgcn_s_barrier();
    }
    };
  if constexpr (H8) {
    // two loops over one skeleton: each kind of k-tile gets its own register allocation (one loop with a run-time switch spilled 517)
    for (int kt = 0; kt < g.h8_t16; ++kt) k_iter(kt, std::false_type{});
    for (int kt = g.h8_t16; kt < nk; ++kt) k_iter(kt, std::true_type{});
  } else {
    for (int kt = 0; kt < nk; ++kt) k_iter(kt, std::false_type{});
  }

  SDN_STAMP(2)
  if constexpr (LNF == 1) {
    // a lane holds the k-chunks fq, fq+4 of rows i*16+fr: the other three lane groups (lane ^ 16, ^ 32) hold the rest
    const float invk = 1.0f / (float)g.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a1 = ln_s1[i], a2 = ln_s2[i];
      a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64);
      a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
      const float mu = a1 * invk;
      const float var = fmaxf(a2 * invk - mu * mu, 0.f);
      ln_mu[i] = mu; ln_rs[i] = __builtin_amdgcn_rsqf(var + g.ln_eps);
    }
  }
  // ---- epilogue ----
  // 16-bit outputs are staged through LDS (free after the k loop) so that HBM sees whole 16-byte-per-lane, row-
  // contiguous stores instead of 8-byte fragments at a row stride (guide T21: "widen the epilogue stores").
  constexpr int CW_PAD = (BN + 8) * 2;                       // staged row stride in bytes (+16 B pad)
  const bool staged = g.out_kind == 0 && g.n_valid == g.N;
  const int out_cols = g.act == 2 ? BN / 2 : BN;             // columns this tile contributes to `out`
  // Residual: its tile is DMA'd into the staging slab first (whole rows, 16 B per lane, every load in flight at once);
  // each lane then adds its own 8-byte slot in fp32 and writes the packed sum back in place.  The DMA image is linear,
  // so that layout has no row pad.  (Per-fragment global loads here cost one serialized L2/HBM round trip each.)
  const bool res_lds = staged && g.res_lds;
  const int CW = res_lds ? BN * 2 : CW_PAD;
  const bool lean = staged && !g.rowgate && g.act == 0 && (res_lds || !g.residual);
  const bool lean_gelu = staged && !g.rowgate && g.act == 3 && !g.residual;
  const bool lean_gate = staged && g.rowgate && g.act == 0 && res_lds;
  const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(g.residual ? g.residual : g.a, g.res_bytes);
  // the staged tile may exceed the LDS (256 x 320): stage PASSES groups of wave-rows one after the other
  constexpr int PASSES = (BM * CW_PAD + 2 * STAGE - 1) / (2 * STAGE);
  constexpr int WM_PER_PASS = WGM / PASSES, ROWS_PER_PASS = 64 * WM_PER_PASS;
  static_assert(WGM % PASSES == 0, "pass split must divide the wave rows");
  static_assert(PASSES <= 2, "epilogue is written for at most two staging passes");
#define SDN_PASS 0
#include "sdn_gemm_epilogue.inc"
#undef SDN_PASS
  if constexpr (PASSES > 1) {
#define SDN_PASS 1
#include "sdn_gemm_epilogue.inc"
#undef SDN_PASS
  }
  SDN_STAMP(3)
#endif  // __HIP_DEVICE_COMPILE__
}

template <typename T, int NREP, int WGM, int NSTAGE, int LNF = 0>
__global__ void __launch_bounds__(128 * WGM, NSTAGE > 2 ? 1 : 2)
k_gemm_dma(const GemmArgs g) { gemm_dma_body<T, NREP, WGM, NSTAGE, LNF, false>(g); }

template <typename T, int NREP>
__global__ void __launch_bounds__(512, 2)
k_gemm_h8(const GemmArgs g) { gemm_dma_body<T, NREP, 4, 2, 0, true>(g); }


// ---- split-K (small M, long K: the 8x8 / 16x16-level convs of a one-prompt batch have 8-20 tiles for 256 CUs and a
// 180..360-tile k loop each).  The k loop is cut into slices (grid.y); every slice writes its fp32 partial tile and this
// kernel sums them in a fixed order (deterministic) and applies the epilogue the unsplit kernel would have applied.
template <typename T>
__global__ void __launch_bounds__(256)
k_splitk_reduce(const float* __restrict__ part, int splits, long slice, int M, int N, const float* __restrict__ bias,
                const float* __restrict__ rowbias, int ld_rowbias, const float* __restrict__ rowgate, int ld_rowgate,
                int rows_per_batch, const unsigned short* __restrict__ residual, int residual_bcast, int ldc, int act,
                unsigned short* __restrict__ out) {
  const int nq = N / 4;
  const long total = (long)M * nq;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int m = (int)(e / nq), n = (int)(e - (long)m * nq) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(part + (long)m * N + n);
    for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(part + (long)s * slice + (long)m * N + n);
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
    const int b = rows_per_batch > 0 ? m / rows_per_batch : 0;
    if (rowbias) v += *reinterpret_cast<const f32x4*>(rowbias + (long)b * ld_rowbias + n);
    if (rowgate) v *= *reinterpret_cast<const f32x4*>(rowgate + (long)b * ld_rowgate + n);
    if (residual) {
      const long rrow = residual_bcast ? (long)(m - b * rows_per_batch) : (long)m;
      const uint2 rr = *reinterpret_cast<const uint2*>(residual + rrow * ldc + n);
      v[0] += T::to_f(rr.x & 0xffff); v[1] += T::to_f(rr.x >> 16); v[2] += T::to_f(rr.y & 0xffff); v[3] += T::to_f(rr.y >> 16);
    }
    if (act == 1) { v[0] = silu_f(v[0]); v[1] = silu_f(v[1]); v[2] = silu_f(v[2]); v[3] = silu_f(v[3]); }
    if (act == 3) { v[0] = gelu_tanh(v[0]); v[1] = gelu_tanh(v[1]); v[2] = gelu_tanh(v[2]); v[3] = gelu_tanh(v[3]); }
    if (act == 4) { v[0] = quick_gelu(v[0]); v[1] = quick_gelu(v[1]); v[2] = quick_gelu(v[2]); v[3] = quick_gelu(v[3]); }
    uint2 pk; pk.x = T::pack2(v[0], v[1]); pk.y = T::pack2(v[2], v[3]);
    *reinterpret_cast<uint2*>(out + (long)m * ldc + n) = pk;
  }
}

}  // namespace sdn_gemm_detail

// Slices worth cutting the k loop into (1 = do not split): only when the tile grid leaves most CUs idle and the k loop
// is long enough for a slice to amortise its prologue; the plan builder sizes the partial buffer from this.
int sdn_gemm_pick_split(int M, int N, int K, int act, int out_kind) {
  if (out_kind != SDN_OUT_BF16 || act == SDN_ACT_GEGLU || (N & 3)) return 1;
  const int nrep = sdn_gemm_pick_tile(M, N, K, act, 0);
  const int bm = nrep >= 8 ? 256 : 128;
  const long tiles = (long)((M + bm - 1) / bm) * (N / (32 * nrep));
  const int nk = K / 64;
  if (tiles >= 96 || nk < 16) return 1;
  int s = (int)(256 / tiles);
  if (s > nk / 8) s = nk / 8;
  if (s > 16) s = 16;
  return s < 2 ? 1 : s;
}

namespace sdn_gemm_detail {

template <typename T, int NREP, int WGM, int NSTAGE = 2, int LNF = 0>
int launch_dma(const GemmArgs& ga, hipStream_t st) {
  const int grid = ga.tiles_m * ga.tiles_n;
  const int splits = ga.kt_per_split > 0 ? (ga.K / BK + ga.kt_per_split - 1) / ga.kt_per_split : 1;
  hipLaunchKernelGGL((k_gemm_dma<T, NREP, WGM, NSTAGE, LNF>), dim3(grid, splits), dim3(128 * WGM), 0, st, ga);
  return sdn_launch_status();
}

template <typename T>
int dispatch_dma(int nrep, const GemmArgs& g, hipStream_t st) {
  if (g.h8_t16 > 0) {                                         // experimental h8 operand form: its own symbol, 256-row tiles only
    if constexpr (T::kDtype == 1) {
      const int grid = g.tiles_m * g.tiles_n;
      if (nrep == 10) hipLaunchKernelGGL((k_gemm_h8<T, 10>), dim3(grid), dim3(512), 0, st, g);
      else if (nrep == 8) hipLaunchKernelGGL((k_gemm_h8<T, 8>), dim3(grid), dim3(512), 0, st, g);
      else return SDN_E_INVALID;
      return sdn_launch_status();
    } else {
      return SDN_E_INVALID;
    }
  }
  if (g.ln_c && g.ln_stats) {                                // LayerNorm-folded form, row statistics from the pre-pass
    switch (nrep) {
      case 10: return launch_dma<T, 10, 4, 2, 2>(g, st);
      case 5: return launch_dma<T, 5, 2, 2, 2>(g, st);
      case 4: return launch_dma<T, 4, 2, 2, 2>(g, st);
      case 2: return launch_dma<T, 2, 2, 2, 2>(g, st);
      default: return SDN_E_INVALID;
    }
  }
  if (g.ln_c) {                                              // ... statistics from the A fragments (narrow N)
    switch (nrep) {
      case 10: return launch_dma<T, 10, 4, 2, 1>(g, st);
      case 5: return launch_dma<T, 5, 2, 2, 1>(g, st);
      case 2: return launch_dma<T, 2, 2, 2, 1>(g, st);
      default: return SDN_E_INVALID;
    }
  }
  switch (nrep) {
    case 10: return launch_dma<T, 10, 4>(g, st);
    case 8: return launch_dma<T, 8, 4>(g, st);
    case 5:
      // at most one workgroup per CU and a long k loop: the deep ring (see NSTAGE) instead of a second resident workgroup
      if (g.tiles_m * g.tiles_n <= 256 && g.K >= 8 * BK && g.kt_per_split == 0 && g_gemm_variant != 5) return launch_dma<T, 5, 2, 4>(g, st);
      return launch_dma<T, 5, 2>(g, st);
    case 4: return launch_dma<T, 4, 2>(g, st);
    case 2: return launch_dma<T, 2, 2>(g, st);
    default: return launch_dma<T, 1, 2>(g, st);
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace sdn_gemm_detail
using namespace sdn_gemm_detail;

// Picks the widest N tile that divides N (160 for SD-v1.4 widths, 128 for GEGLU / MMDiT widths, 64, 32).
int sdn_gemm_pick_nrep(int n_padded, int act) {
  if (act == SDN_ACT_GEGLU) return (n_padded % 128 == 0) ? 4 : ((n_padded % 64 == 0) ? 2 : 0);
  if (n_padded % 160 == 0) return 5;
  if (n_padded % 128 == 0) return 4;
  if (n_padded % 64 == 0) return 2;
  if (n_padded % 32 == 0) return 1;
  return 0;
}

// Undeclared debug hook for in-process A/B timing (tools/bench_gemm.py): 2 = tile heuristic off, >=16: ablations.
extern "C" void sdn_debug_set_gemm_variant(int v) { g_gemm_variant = v; }
// diagnostics (tests, bench.py): how many sdn_gemm_* calls since the last reset went to the slab-ring convolution kernel
// (out[0]) and how many to the implicit-GEMM / plain tile (out[1]) -- so that a test of the slab path cannot pass vacuously
// and a plan label can be checked against what was actually launched
static long long g_gemm_launches[2] = {0, 0};
extern "C" void sdn_debug_gemm_launch_counts(long long* out2, int reset) {
  if (out2) { out2[0] = g_gemm_launches[0]; out2[1] = g_gemm_launches[1]; }
  if (reset) g_gemm_launches[0] = g_gemm_launches[1] = 0;
}
extern "C" void sdn_debug_set_gemm_stamps(void* p) { g_gemm_stamps = (unsigned long long*)p; }

// Tile choice with the grid in mind: when the widest tile leaves the 256 CUs (x2 resident blocks) underfilled
// (the 8x8 / 16x16 levels at small batch), fall back to BN = 64 to multiply the number of workgroups.
int sdn_gemm_pick_tile(int M, int N, int K, int act, int epilogue_reads) {
  int nrep = sdn_gemm_pick_nrep(N, act);
  if (g_gemm_variant == 2) return nrep;                      // debug: heuristics off
  // big tile (256 rows, 8 waves, 1 block/CU) when it still fills the chip: >= ~3/4 of the 256 CUs get a tile
  if (g_gemm_variant != 3) {
    int big = (N % 320 == 0) ? 10 : ((N % 256 == 0) ? 8 : 0);
    // N divisible by both (1280): one workgroup per CU runs whole waves of 256 tiles, so a grid of 192 tiles of 256 x 320 (the
    // 8x8-level convs at 3 x 64 samples: M = 12288, N = 1280) leaves a quarter of the chip idle for the whole launch, while the
    // same problem as 240 tiles of 256 x 256 fills 94 % of it with tiles 20 % shorter.  Cost = waves x tile width; ties keep 320.
    if (big == 10 && N % 256 == 0 && act == SDN_ACT_NONE && g_gemm_variant != 14) {                     // variant 14: rule off (A/B)
      const long tm = (M + 255) / 256;
      const long w10 = (tm * (N / 320) + 255) / 256 * 320, w8 = (tm * (N / 256) + 255) / 256 * 256;
      if (w8 < w10) big = 8;
    }
    // Round 1 kept short k loops (K = 320 .. 1280) on the 2-blocks-per-CU tile: with one block per CU nothing hides a tile's
    // epilogue, and that epilogue was then as long as the k loop.  After the lean epilogues and the cheaper GELU the picture
    // (tools/bench_gemm.py, round 2, B = 128) is: the big tile WINS 17-30 % on every short-K shape whose epilogue only computes
    // and stores (qkv, proj_in, GEGLU at K >= 640: half the L2 -> LDS bytes per FLOP) and LOSES 5-17 % where the epilogue has
    // to fetch a residual tile first (two staging passes, each an exposed HBM round trip) -- those stay on the small tile
    // until the k loop is long enough to amortise it.
    const bool long_k = K >= 2048 || (K >= 1280 && act == SDN_ACT_GEGLU) || (K >= 1536 && (act == SDN_ACT_NONE || act == SDN_ACT_GELU_TANH));
    const bool short_ok = !epilogue_reads && K >= 320 && g_gemm_variant != 9;   // variant 9: round-1 rule (A/B)
    if (big && (long_k || short_ok)) {
      const long tiles = (long)((M + 255) / 256) * (N / (32 * big));
      if (tiles >= 192) return big;
    }
  }
  const int tiles_m = (M + 127) / 128;
  if (nrep > 2 && N % 64 == 0 && tiles_m * (N / (32 * nrep)) < 256) nrep = 2;
  return nrep;
}

static int sdn_gemm_impl(int dtype, const sdn_gemm_desc* d, const void* a, const void* a2, const void* w,
                         const float* bias, const float* rowbias, const float* rowgate, const void* residual, void* out,
                         void* stream, void* partials = nullptr, size_t partial_bytes = 0, const float* ln_c = nullptr,
                         const float* ln_d = nullptr, float ln_eps = 0.f, const float* ln_stats = nullptr, float* col_stats = nullptr) {
  if (!d || !a || !w || !out) return SDN_E_INVALID;
  if (d->M < 0 || d->N <= 0 || d->K <= 0 || (d->K % BK) != 0) return SDN_E_INVALID;
  if (d->M == 0) return SDN_OK;
  const int n_valid = d->n_valid > 0 ? d->n_valid : d->N;
  if (n_valid > d->N) return SDN_E_INVALID;
  if (sdn_gemm_pick_nrep(d->N, d->act) == 0) return SDN_E_INVALID;
  const int x3 = d->x3_out;
  const bool h8 = x3 == 5;                                  // experimental: fp16 + e4m3 corrections by operand expansion
  if (x3 < 0 || x3 > 5 || (x3 && ((dtype != 0) != h8 || rowgate || d->split_k > 1 || partials || ln_c || col_stats || n_valid != d->N)))
    return SDN_E_INVALID;
  if (h8 && (d->a_mode != SDN_A_PLAIN || d->act != SDN_ACT_NONE || (d->K % 256) != 0 || d->res_pre || (d->K1 > 0 && d->K1 < d->K)))
    return SDN_E_INVALID;
  if (x3 && ((x3 == 2) != (d->act == SDN_ACT_GEGLU) || (x3 != 2 && d->act != SDN_ACT_NONE) || (residual && !al16(residual)) || !al16(out)))
    return SDN_E_INVALID;
  // (in the bf16x3 plan the residual rides in the accumulators from the start: it is not an epilogue read)
  // res_pre: the 16-bit residual goes into the accumulators before the k loop (plain k_gemm_dma tiles only: not split-K, not the
  // LayerNorm-folded or column-sum forms, whole-width 16-bit output)
  const bool res_pre = d->res_pre && residual && !x3 && !rowgate && !d->residual_bcast && d->split_k <= 1 && !partials && !ln_c &&
                       d->out_kind == SDN_OUT_BF16 && n_valid == d->N && d->act == SDN_ACT_NONE;
  int nrep = sdn_gemm_pick_tile(d->M, d->N, d->K, d->act, ((residual && !x3 && !res_pre) || rowgate) ? 1 : 0);
  if ((ln_c || ln_d) && nrep == 8 && d->N % 320 == 0) nrep = 10;          // (the LayerNorm-folded forms have no 256-wide instantiation)
  if (h8) {                                                                // the fp8 k-tiles exist on the 256-row tiles only; the 256-wide
    if (d->N % 256 == 0) nrep = 8;                                         // one holds its fp8 body in registers (the 320-wide one spills
    else if (nrep < 8) return SDN_E_INVALID;                               // ~100 of them: measured, slower), so it is taken where it divides N
  }
  if (!al16(a) || !al16(w) || (a2 && !al16(a2)) || (residual && (reinterpret_cast<uintptr_t>(residual) & 7)) ||
      (reinterpret_cast<uintptr_t>(out) & 7) || (bias && !al16(bias)) || (rowbias && !al16(rowbias)))
    return SDN_E_INVALID;
  GemmArgs g{};
  g.a = (const __bf16*)a; g.a2 = (const __bf16*)a2; g.w = (const __bf16*)w;
  g.bias = bias; g.rowbias = rowbias; g.rowgate = rowgate; g.residual = res_pre ? nullptr : (const __bf16*)residual; g.out = out;
  g.res_pre = res_pre ? (const __bf16*)residual : nullptr;
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
    if (d->asym_pad != 0 && (d->asym_pad != 1 || d->stride != 2 || d->upsample)) return SDN_E_INVALID;
    const int pad2 = d->asym_pad ? 1 : 2;                     // total zero padding per axis
    if ((Hi + pad2 - 3) / d->stride + 1 != d->Ho || (Wi + pad2 - 3) / d->stride + 1 != d->Wo) return SDN_E_INVALID;
    g.K1 = d->K; g.Hs = d->Hs; g.Ws = d->Ws; g.Cin = d->Cin; g.Ho = d->Ho; g.Wo = d->Wo; g.stride = d->stride;
    g.upsample = d->upsample; g.conv_off = d->asym_pad ? 1 : 0;
  } else {
    return SDN_E_INVALID;
  }
  if (d->act < 0 || d->act > 4 || d->out_kind < 0 || d->out_kind > 2) return SDN_E_INVALID;
  if (d->act == SDN_ACT_GEGLU && ((d->out_kind != SDN_OUT_BF16 && !x3) || rowbias || rowgate || residual || n_valid != d->N))
    return SDN_E_INVALID;
  if ((rowbias || rowgate || d->residual_bcast || d->out_kind == SDN_OUT_F32_NCHW) && d->rows_per_batch <= 0) return SDN_E_INVALID;
  if (rowgate && !al16(rowgate)) return SDN_E_INVALID;
  g.x3_out = h8 ? 1 : x3;                                    // (h8: f32 rows straight from the accumulators, as x3_out = 1)
  g.h8_t16 = h8 ? d->K / 128 : 0;                            // d->K counts 16-bit units of the 4-byte-per-element row: half of it is fp16
  g.act = d->act; g.out_kind = x3 ? SDN_OUT_F32 : d->out_kind; g.rows_per_batch = d->rows_per_batch; g.ld_rowbias = d->ld_rowbias;
  g.ld_rowgate = d->ld_rowgate; g.residual_bcast = d->residual_bcast;
  g.n_valid = n_valid;
  g.ldc = d->ldc > 0 ? d->ldc : (d->act == SDN_ACT_GEGLU ? d->N / 2 : n_valid);
  if (!x3 && d->out_kind == SDN_OUT_BF16 && ((g.ldc & 7) || !al16(out))) return SDN_E_INVALID;
  if (x3 && (g.ldc & 3)) return SDN_E_INVALID;
  const int bn = 32 * nrep;
  const int bm = nrep >= 8 ? 256 : 128;
  g.tiles_m = (d->M + bm - 1) / bm; g.tiles_n = d->N / bn;
  {
    const long res_rows = d->residual_bcast ? (long)d->rows_per_batch : (long)d->M;
    const long res_bytes = res_rows * g.ldc * 2;
    g.res_lds = !x3 && !res_pre && residual != nullptr && al16(residual) && res_bytes < (1L << 31) && g_gemm_variant != 4;   // variant 4: per-fragment loads (A/B)
    g.res_bytes = g.res_lds ? (unsigned)res_bytes : 0u;
  }
  if (col_stats) {                                              // column statistics ride on the staged 16-bit tile
    if (d->out_kind != SDN_OUT_BF16 || n_valid != d->N || d->act == SDN_ACT_GEGLU || d->split_k > 1 ||
        (reinterpret_cast<uintptr_t>(col_stats) & 7))
      return SDN_E_INVALID;
    g.col_stats = col_stats;
  }
  if (ln_c || ln_d) {                                          // LayerNorm-folded form (sdn_gemm_ln_*)
    if (!ln_c || !ln_d || !al16(ln_c) || !al16(ln_d) || d->a_mode != SDN_A_PLAIN || g.K1 != d->K || bias || rowbias || rowgate ||
        residual || d->out_kind != SDN_OUT_BF16 || n_valid != d->N || d->split_k > 1 ||
        (d->act != SDN_ACT_NONE && d->act != SDN_ACT_GEGLU) || (ln_stats && (reinterpret_cast<uintptr_t>(ln_stats) & 7)) ||
        (ln_stats ? (nrep != 10 && nrep != 5 && nrep != 4 && nrep != 2) : (nrep != 10 && nrep != 5 && nrep != 2)))
      return SDN_E_INVALID;
    g.ln_c = ln_c; g.ln_d = ln_d; g.ln_eps = ln_eps; g.ln_stats = ln_stats;
  }
  g.panel = (g.tiles_n > 4 && g_gemm_variant != 15) ? 4 : 0;                // variant 15: row-major tile order (A/B)
  g.dbg = g_gemm_variant >= 16 ? (g_gemm_variant >> 4) : 0;
  g.stamps = g_gemm_stamps;
  hipStream_t st = (hipStream_t)stream;
  // operands must stay below the LDS-DMA out-of-range sentinel (2 GiB per tensor)
  const long a_rows = d->a_mode == SDN_A_CONV3X3 ? (long)(d->M / (d->Ho * d->Wo)) * d->Hs * d->Ws : (long)d->M;
  const long a_ld1 = d->a_mode == SDN_A_CONV3X3 ? d->Cin : g.K1, a_ld2 = d->a_mode == SDN_A_CONV3X3 ? 0 : d->K - g.K1;   // per SOURCE
  if (a_rows * a_ld1 * 2 >= (1L << 31) || a_rows * a_ld2 * 2 >= (1L << 31) || (long)d->N * d->K * 2 >= (1L << 31))
    return SDN_E_INVALID;
  if (d->split_k > 1) {
    // split-K: fp32 partials per k slice, then one deterministic reduce + epilogue pass
    const int nk = d->K / BK;
    if (!partials || d->out_kind != SDN_OUT_BF16 || d->act == SDN_ACT_GEGLU || n_valid != d->N || (d->N & 3) || d->split_k > nk ||
        (reinterpret_cast<uintptr_t>(partials) & 15))
      return SDN_E_INVALID;
    const int per = (nk + d->split_k - 1) / d->split_k, splits = (nk + per - 1) / per;
    const long slice = (long)d->M * d->N;
    if (partial_bytes < (size_t)splits * slice * 4) return SDN_E_WORKSPACE;
    GemmArgs gp = g;
    gp.bias = nullptr; gp.rowbias = nullptr; gp.rowgate = nullptr; gp.residual = nullptr; gp.res_lds = 0; gp.res_bytes = 0;
    gp.act = 0; gp.out_kind = SDN_OUT_F32; gp.out = partials; gp.ldc = d->N; gp.n_valid = d->N;
    gp.kt_per_split = per; gp.split_stride = slice * 4;
    const int rc = dtype == 0 ? dispatch_dma<SdnBF16>(nrep, gp, st) : dispatch_dma<SdnF16>(nrep, gp, st);
    if (rc != SDN_OK) return rc;
    const long total = slice / 4;
    long grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (dtype == 0)
      hipLaunchKernelGGL((k_splitk_reduce<SdnBF16>), dim3((unsigned)grid), dim3(256), 0, st, (const float*)partials, splits, slice,
                         d->M, d->N, bias, rowbias, d->ld_rowbias, rowgate, d->ld_rowgate, d->rows_per_batch,
                         (const unsigned short*)residual, d->residual_bcast, g.ldc, d->act, (unsigned short*)out);
    else
      hipLaunchKernelGGL((k_splitk_reduce<SdnF16>), dim3((unsigned)grid), dim3(256), 0, st, (const float*)partials, splits, slice,
                         d->M, d->N, bias, rowbias, d->ld_rowbias, rowgate, d->ld_rowgate, d->rows_per_batch,
                         (const unsigned short*)residual, d->residual_bcast, g.ldc, d->act, (unsigned short*)out);
    return sdn_launch_status();
  }
  if (nrep == 10 && g_gemm_variant != 13) {                    // variant 13: slab convolution off (A/B, equality tests)
    const int rc = dispatch_conv_slab(dtype, g, st);
    if (rc != SDN_GEMM_NOT_SLAB) { ++g_gemm_launches[0]; return rc; }
  }
  ++g_gemm_launches[1];
  return dtype == 0 ? dispatch_dma<SdnBF16>(nrep, g, st) : dispatch_dma<SdnF16>(nrep, g, st);
}

extern "C" int sdn_gemm_bf16(const sdn_gemm_desc* d, const void* a, const void* a2, const void* w, const float* bias,
                             const float* rowbias, const float* rowgate, const void* residual, void* out, void* stream) {
  return sdn_gemm_impl(0, d, a, a2, w, bias, rowbias, rowgate, residual, out, stream);
}
extern "C" int sdn_gemm_f16(const sdn_gemm_desc* d, const void* a, const void* a2, const void* w, const float* bias,
                            const float* rowbias, const float* rowgate, const void* residual, void* out, void* stream) {
  return sdn_gemm_impl(1, d, a, a2, w, bias, rowbias, rowgate, residual, out, stream);
}

// GEMM that also emits, per block of 128 output rows and per column, the (sum, sum of squares) of the 16-bit values it
// stores: col_stats [ceil(M / 128)][N][2] f32.  The GroupNorm over this tensor (sdn_groupnorm_cols_*) then needs no pass of
// its own for the statistics.  16-bit output, n_valid == N, no GEGLU.
extern "C" int sdn_gemm_stats_bf16(const sdn_gemm_desc* d, const void* a, const void* a2, const void* w, const float* bias,
                                   const float* rowbias, const void* residual, void* out, float* col_stats, void* stream) {
  if (!col_stats) return SDN_E_INVALID;
  return sdn_gemm_impl(0, d, a, a2, w, bias, rowbias, nullptr, residual, out, stream, nullptr, 0, nullptr, nullptr, 0.f, nullptr, col_stats);
}
extern "C" int sdn_gemm_stats_f16(const sdn_gemm_desc* d, const void* a, const void* a2, const void* w, const float* bias,
                                  const float* rowbias, const void* residual, void* out, float* col_stats, void* stream) {
  if (!col_stats) return SDN_E_INVALID;
  return sdn_gemm_impl(1, d, a, a2, w, bias, rowbias, nullptr, residual, out, stream, nullptr, 0, nullptr, nullptr, 0.f, nullptr, col_stats);
}

extern "C" int sdn_gemm_splitk_bf16(const sdn_gemm_desc* d, const void* a, const void* a2, const void* w, const float* bias,
                                    const float* rowbias, const float* rowgate, const void* residual, void* out, void* partials,
                                    size_t partial_bytes, void* stream) {
  return sdn_gemm_impl(0, d, a, a2, w, bias, rowbias, rowgate, residual, out, stream, partials, partial_bytes);
}
extern "C" int sdn_gemm_splitk_f16(const sdn_gemm_desc* d, const void* a, const void* a2, const void* w, const float* bias,
                                   const float* rowbias, const float* rowgate, const void* residual, void* out, void* partials,
                                   size_t partial_bytes, void* stream) {
  return sdn_gemm_impl(1, d, a, a2, w, bias, rowbias, rowgate, residual, out, stream, partials, partial_bytes);
}

// LayerNorm folded into the GEMM: see k_gemm_dma (LNF).  w_folded = W * gamma per input channel (16 bit), c[n] = sum_k
// w_folded[n, k], d[n] = sum_k beta[k] W[n, k] + bias[n]  (produced once per weight set by sdn_ln_fold_*).
extern "C" int sdn_gemm_ln_bf16(const sdn_gemm_desc* d, const void* a, const void* w_folded, const float* c, const float* dvec,
                                float eps, const float* row_stats, void* out, void* stream) {
  return sdn_gemm_impl(0, d, a, nullptr, w_folded, nullptr, nullptr, nullptr, nullptr, out, stream, nullptr, 0, c, dvec, eps, row_stats);
}
extern "C" int sdn_gemm_ln_f16(const sdn_gemm_desc* d, const void* a, const void* w_folded, const float* c, const float* dvec,
                               float eps, const float* row_stats, void* out, void* stream) {
  return sdn_gemm_impl(1, d, a, nullptr, w_folded, nullptr, nullptr, nullptr, nullptr, out, stream, nullptr, 0, c, dvec, eps, row_stats);
}

// ---- the fold itself: one workgroup per output row n ----
namespace {
template <typename T>
__global__ void __launch_bounds__(256)
k_ln_fold(const unsigned short* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
          const float* __restrict__ bias, int K, unsigned short* __restrict__ wf, float* __restrict__ c, float* __restrict__ dvec) {
  __shared__ float red[8];
  const long n = blockIdx.x;
  float sc = 0.f, sd = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float wv = T::to_f(w[n * K + k]);
    const unsigned r = T::pack2(wv * gamma[k], 0.f) & 0xffffu;        // the rounded value is what the MFMA will see
    wf[n * K + k] = (unsigned short)r;
    sc += T::to_f(r);
    sd = fmaf(beta[k], wv, sd);
  }
  sc = block_sum<4>(sc, red);
  sd = block_sum<4>(sd, red + 4);
  if (threadIdx.x == 0) { c[n] = sc; dvec[n] = sd + (bias ? bias[n] : 0.f); }
}
}  // namespace

extern "C" int sdn_ln_fold(int32_t dtype, const void* w, const float* gamma, const float* beta, const float* bias, int32_t rows,
                           int32_t cols, void* w_folded, float* c, float* dvec, void* stream) {
  if (!w || !gamma || !beta || !w_folded || !c || !dvec || rows <= 0 || cols <= 0 || dtype < 0 || dtype > 1) return SDN_E_INVALID;
  if (dtype == 1)
    hipLaunchKernelGGL((k_ln_fold<SdnF16>), dim3(rows), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)w, gamma, beta,
                       bias, cols, (unsigned short*)w_folded, c, dvec);
  else
    hipLaunchKernelGGL((k_ln_fold<SdnBF16>), dim3(rows), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)w, gamma, beta,
                       bias, cols, (unsigned short*)w_folded, c, dvec);
  return sdn_launch_status();
}

// ---- product of two linears (sdn_linear_pair_fold, sdn_ops.h): prepare-time, one thread per output element --------------
namespace {
template <typename T>
__global__ void __launch_bounds__(256)
k_linear_pair_fold(const unsigned short* __restrict__ wa, const unsigned short* __restrict__ wb, const float* __restrict__ ba,
                   const float* __restrict__ bb, int C, int K, unsigned short* __restrict__ wcat, float* __restrict__ bcat) {
  const int n = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int KC = K + C;
  if (k < K) {
    float acc = 0.f;
    for (int j = 0; j < C; ++j) acc = fmaf(T::to_f(wb[(long)n * C + j]), T::to_f(wa[(long)j * K + k]), acc);
    wcat[(long)n * KC + k] = (unsigned short)(T::pack2(acc, 0.f) & 0xffffu);
  } else if (k < KC) {
    wcat[(long)n * KC + k] = wb[(long)n * C + (k - K)];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float acc = bb ? bb[n] : 0.f;
    if (ba) for (int j = 0; j < C; ++j) acc = fmaf(T::to_f(wb[(long)n * C + j]), ba[j], acc);
    bcat[n] = acc;
  }
}
}  // namespace

int sdn_linear_pair_fold(int dtype, const void* wa, const void* wb, const float* ba, const float* bb, int C, int K, void* w_cat,
                         float* b_cat, void* stream) {
  if (!wa || !wb || !w_cat || !b_cat || C <= 0 || K <= 0 || dtype < 0 || dtype > 1) return SDN_E_INVALID;
  const dim3 grid((K + C + 255) / 256, C);
  if (dtype == 1)
    hipLaunchKernelGGL((k_linear_pair_fold<SdnF16>), grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)wa,
                       (const unsigned short*)wb, ba, bb, C, K, (unsigned short*)w_cat, b_cat);
  else
    hipLaunchKernelGGL((k_linear_pair_fold<SdnBF16>), grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)wa,
                       (const unsigned short*)wb, ba, bb, C, K, (unsigned short*)w_cat, b_cat);
  return sdn_launch_status();
}
