// Fused attention forward (row U6): softmax(Q K^T * scale) V, flash-style, 16-bit in / fp32 accumulate,
// v_mfma_f32_32x32x16.  Head dims 40/80/160 (SD-v1.4: 8 heads at C=320/640/1280) and 64 (MMDiT).
//
// Orientation (CDNA4-specific, see the guide's "accumulator tile as the next MFMA's operand"):
//   S^T = K . Q^T      A = K rows from LDS (ds_read_b128), B = Q rows held in registers for the whole kernel
//                      -> a lane owns ONE query column (lane & 31) and 16 keys per 32-key block in registers,
//                         so the softmax max is a per-lane loop + one v_permlane32_swap;
//   O^T += V^T . P^T   B = the S^T accumulator itself, converted to 16 bit in registers (no LDS round trip),
//                      A = V^T read from the row-major V tile with ds_read_b64_tr_b16 (hardware transpose).
// Small head dims make this kernel VALU-bound (the softmax costs the same per score whatever d is), so the
// per-score instruction count is what is optimised:
//   * p = v_exp_f32(fma(s, c, -m c))                      2 VALU per score (scale folded into the exponent);
//   * the row sum comes out of the MFMA: the zero padding of the V tile (40 -> 64, 80 -> 96 columns) holds a
//     column of ones, so row `HD` of O^T accumulates sum_k p -- no adds, and it is rescaled with O for free;
//   * O is rescaled only when some lane's running max moved (wave-uniform skip);
//   * cross-half max via v_permlane32_swap instead of an LDS permute.
// K/V tiles are double-buffered in LDS through registers: the global loads of tile t+1 are issued before the
// MFMAs of tile t and written after them (one barrier per tile).  Head dim is zero-padded inside LDS only; HBM
// tensors stay packed [B, N, heads*d], i.e. the NHWC token matrix the projections write.
#include <stdlib.h>

#include <type_traits>

#include "sdn_common.h"
#include "sdn_ops.h"

namespace sdn_attn_detail {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#if defined(__HIP_DEVICE_COMPILE__)     // buffer-resource builtins exist only in the device pass
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#endif

// Diagnostics build (-DSDN_ATTN_STAMPS, tools/attn_stamps.py): per-wave cycle sums of the main loop's phases.
#ifdef SDN_ATTN_STAMPS
#define SDN_ATS_DECL unsigned long long ats_prev = 0, ats_sum[7] = {0, 0, 0, 0, 0, 0, 0};
#define SDN_ATS_T0 unsigned long long ats_t0; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ats_t0)::"memory");
#define SDN_ATS_MARK(I)                                                                         \
  {                                                                                             \
    unsigned long long t_;                                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
    if ((I) >= 0) ats_sum[(I) < 0 ? 0 : (I)] += t_ - ats_prev;                                  \
    ats_prev = t_;                                                                              \
  }
#define SDN_ATS_FLUSH                                                                           \
  if (g_attn_stamps && lane == 0)                                                               \
    for (int i_ = 0; i_ < 7; ++i_) g_attn_stamps[((long)blockIdx.x * 4 + wid) * 7 + i_] = ats_sum[i_];
__device__ unsigned long long* g_attn_stamps = nullptr;
#define SDN_ATS_PRO ats_sum[5] = ats_prev - ats_t0;       /* kernel start -> main loop (Q load, LDS init, first K/V tile) */
#else
#define SDN_ATS_DECL
#define SDN_ATS_T0
#define SDN_ATS_PRO
#define SDN_ATS_MARK(I)
#define SDN_ATS_FLUSH
#endif

constexpr int THREADS = 256;
[[maybe_unused]] constexpr int KV = 64;       // keys per tile
constexpr int QB = 128;      // queries per workgroup (32 per wave)

struct AttnArgs {
  const unsigned short* q; const unsigned short* k; const unsigned short* v; unsigned short* out;
  int nq, nk, ldq, ldk, ldv, ldo;
  float c;                   // scale * log2(e)
  int heads, nqb, npairs;    // q-blocks per (batch, head) pair; number of pairs
  // optional second segment (MMDiT joint attention: rows [0,n1) = image tokens, [n1, n) = text tokens, each stream
  // in its own buffer -> the concatenated sequence is never materialised)
  const unsigned short* q2; const unsigned short* k2; const unsigned short* v2; unsigned short* out2;
  int n1, ldq2, ldk2, ldv2, ldo2;
  int head_inner;                    // block order for short key sets, see the block-index decode
  int causal; const int* kmask;      // MASK instantiation only: key <= query; kmask [B, nk] (0 = padded key), nullable
  int nomax;                         // bf16: first pass without the per-tile running maximum (see "optimistic pass"); 0 = guarded pass only
};

// Row `row` of sample b in a (possibly two-segment) [B, n, ld] tensor.
template <bool SEG>
__device__ __forceinline__ const unsigned short* row_ptr(const unsigned short* p1, const unsigned short* p2, int ld1,
                                                          int ld2, int n1, int n, int b, int row) {
  if (SEG && row >= n1) return p2 + ((long)b * (n - n1) + (row - n1)) * ld2;
  return p1 + ((long)b * (SEG ? n1 : n) + row) * ld1;
}

// MASK (CLIP text encoder): causal mask (key <= query) and an optional per-sample key-padding mask; a separate
// instantiation so that the UNet / MMDiT kernels carry no trace of it.
// QS = query sets per wave (32 queries each).  QS = 2 (d = 40 self-attention): every K / V fragment read from LDS feeds TWO
// MFMAs, halving the LDS bytes per FLOP, and the softmax VALU of one set can issue under the MFMAs of the other inside ONE
// wave -- the d = 40 loop is bound by vector issue + LDS + matrix pipe all at ~60 %, at a clock the chip holds
// down; less LDS traffic per MFMA is the lever that raises it.  Costs registers: 2 waves per SIMD instead of 4.
template <typename T, int HD, bool SEG, bool MASK = false, bool DMA = !SEG, int QS = 1>
__global__ void __launch_bounds__(THREADS)
k_attn(const AttnArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  SDN_ATS_T0
  constexpr int KQ = (HD + 15) / 16;          // 16-deep k-steps of Q K^T
  constexpr int NDB = (HD + 31) / 32;         // 32-row blocks of O^T
  constexpr bool ONES = (HD % 32) != 0;       // a free padding column exists -> row sums via MFMA
  constexpr bool OFFSET_FREE = T::kDtype == 0; // bf16 has fp32's exponent range: softmax without max subtraction (below)
  // fp16 with a FREE padding column in the Q K^T contraction (d = 40 -> 48) and LDS-DMA tiles -- round 5.  fp16's P needs p < 2^16,
  // so the bf16 form (no offset at all) is out; the classic form pays one FMA + 1/2 v_max3 per score that the bf16 loop does not
  // (VALU is this kernel's bound: 31.4 vs 23.5 ms per forward at B = 192).  Here the scale is folded into Q as for bf16 and the
  // query's exponent is CENTRED on its first tile's maximum m0 once: column HD of every K row reads 1.0 (the static block) and
  // column HD of the Q fragment is set to -m0 after tile 0, so the MFMA itself delivers s - m0 for all later tiles -- no per-score
  // subtraction, no running maximum.  p <= 2^15 is checked afterwards through the row sum (sum >= p); a workgroup with a query
  // outside [2^-10, 2^15] re-runs its block with the guarded (running-maximum) loop.
  constexpr bool CENTRED = T::kDtype == 1 && !SEG && !MASK && DMA && (KQ * 16 > HD);
  // The other unmasked fp16 instances (d = 64 / 80 / 160: no free column) take the same idea with the subtraction spelled out:
  // p = 2^(s - m0), one v_sub per score (what the classic form's FMA cost) but no running maximum (16 v_max3, a cross-half swap, two
  // ballots and the conditional rescale per query set and tile), checked and re-run the same way.  MMDiT's joint attention
  // (d = 64, 18-38 % of the SD-v3 forward) is bound by vector issue: 640 vs 512 cycles per 64-key tile and query set.
  constexpr bool CENTRED_SUB = T::kDtype == 1 && !MASK && !CENTRED;
  constexpr bool QSCALED = OFFSET_FREE || CENTRED || CENTRED_SUB;
  static_assert(!CENTRED || ((HD % 32) != 0 && HD - 16 * (KQ - 1) == 8), "centred form: row sums by the ones column, column HD = element 0 of the h = 1 lanes");
  constexpr int CH = HD / 8;                  // valid 16-B chunks per K / V row
  // Single-buffer inputs (DMA): K/V tiles arrive by LDS-DMA (buffer_load ... lds, 16 B per lane, one linear 1-KiB
  // piece per wave-instruction): no staging registers, no ds_write pass.  A DMA piece cannot skip bytes, so the tiles
  // hold ONLY the HD data columns; the zero padding of the head dim and the ones column live once in a 32-byte static
  // block that the padding lanes of the fragment reads point at instead (their contents do not depend on the key).
  // Two-segment inputs (SEG, MMDiT): the streams have different base pointers, so DMA needs every 64-key tile to lie in
  // ONE stream (n1 % 64 == 0) and both streams to share the row strides; otherwise the register-staged path runs.
#ifndef SDN_ATTN_VCH40
#define SDN_ATTN_VCH40 5
#endif
  constexpr int KCH = HD == 40 ? 5 : CH + 1;  // 16-B chunks per K row in LDS; strides 20/36/44/84 words: b128 reads conflict-free
  constexpr int VCH = HD == 40 ? SDN_ATTN_VCH40 : (HD == 160 ? 20 : 12);   // V rows 16 or 48 banks apart (tr reads conflict-free)
  // K tile row stride in bytes (register staging: zero-padded to KQ*16 columns + 16 B against bank conflicts)
  constexpr int KSTR = DMA ? KCH * 16 : KQ * 32 + 16;
  // V tile row stride in bytes.  ds_read_b64_tr_b16 is banked like ds_read_b64 (64 banks, lanes 0-31 / 32-63 are the
  // two groups): one group reads 4 consecutive keys x 64 contiguous bytes, so the rows must sit 16 or 48 banks apart
  // for the 4 x 16 words to tile all 64 banks.
  constexpr int VSTR_PAD = (NDB * 16) % 64 == 16 || (NDB * 16) % 64 == 48 ? NDB * 64 : NDB * 64 + 64;
  constexpr int VSTR = DMA ? VCH * 16 : VSTR_PAD;
  constexpr int STAGE = KV * KSTR + KV * VSTR;
  constexpr int ZOFF = 2 * STAGE;             // static block: [1,0,0,0 | 0,0,0,0 | 0 x 8] (16-bit), DMA mode only
  constexpr int NLOAD = (2 * KV * CH + THREADS - 1) / THREADS;   // staged chunks per thread per tile (register staging)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE + 48];          // + static block (32 B) + the re-run flag

  // K/V tile DMA of the next tile: at the top of the iteration, or (d = 40: +1.5-3 %; d = 80 / 160 lose 3-6 %) after the QK^T MFMAs
  constexpr bool kLateDma = HD == 40;
#ifdef SDN_ATTN_NO_ASM_TR
  constexpr bool kAsmTrOn = false;             // A/B build: the builtin tr reads (and the compiler's mid-loop vmcnt(0))
#else
  constexpr bool kAsmTrOn = true;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: LDS-DMA bases go to M0
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware mapping (1-D grid): workgroups are dealt round-robin over the 8 XCDs, so id % 8 labels the XCD.
  // All q-blocks of one (batch, head) pair go to ONE XCD: its K/V (N*d*4 B = 655 KB at N=4096, d=40) is then fetched
  // into that XCD's 4 MiB L2 once and re-read from there by the other q-blocks, instead of thrashing all 8 L2s.
  int pair, qblk;
  {
    const int nqb = a.nqb, npairs = a.npairs, id = blockIdx.x;
    if (a.head_inner) {
      // short key sets (cross-attention): K/V locality is irrelevant (77 rows), what matters is that the 8 heads of one
      // query block -- 80-byte slices of the SAME 640-byte Q / O rows -- run back to back on ONE XCD, so those lines are
      // fetched into / merged in one L2 instead of eight.  XCD x takes the samples b = x (mod 8); heads innermost.
      const int x = id & 7, j = id >> 3;
      const int hh = j % a.heads, rest = j / a.heads;
      qblk = rest % nqb;
      pair = ((rest / nqb) * 8 + x) * a.heads + hh;
    } else if ((npairs & 7) == 0) {
      const int x = id & 7, j = id >> 3;
      pair = (j / nqb) * 8 + x; qblk = j - (j / nqb) * nqb;
    } else {
      pair = id / nqb; qblk = id - pair * nqb;
    }
  }
  const int head = pair % a.heads, b = pair / a.heads;
  const int q0 = qblk * (QB * QS) + wid * (32 * QS);
  bool qvalid[QS];                               // per lane: the query tail (nq % 32 != 0) is clamped, not stored
#pragma unroll
  for (int qs = 0; qs < QS; ++qs) qvalid[qs] = q0 + 32 * qs + r < a.nq;

  // ---- one-time LDS init: zero the padding columns, plant the ones column (both stages) ----
  if (tid == 8) *reinterpret_cast<unsigned*>(smem + ZOFF + 32) = 0u;          // re-run flag of the optimistic pass
  if constexpr (DMA) {
    if (tid < 8) *reinterpret_cast<unsigned*>(smem + ZOFF + tid * 4) = tid == 0 ? (T::pack2(1.0f, 0.0f) & 0xffffu) : 0u;
  } else {
    const unsigned one16 = T::pack2(1.0f, 0.0f) & 0xffffu;
    for (int e = tid; e < 2 * KV; e += THREADS) {
      unsigned char* sK = smem + (e / KV) * STAGE + (e % KV) * KSTR;
      unsigned char* sV = smem + (e / KV) * STAGE + KV * KSTR + (e % KV) * VSTR;
      for (int c = CH * 16; c < KQ * 32; c += 4) *reinterpret_cast<unsigned*>(sK + c) = 0u;
      for (int c = CH * 16; c < NDB * 64; c += 4) *reinterpret_cast<unsigned*>(sV + c) = (ONES && c == CH * 16) ? one16 : 0u;
    }
  }

  // ---- Q fragments (B operand: lane (r,h) holds Q[q0+r][16s + 8h .. +8)) ----
  typename T::v8 qf[QS][KQ];
#pragma unroll
  for (int qs = 0; qs < QS; ++qs)
#pragma unroll
  for (int s = 0; s < KQ; ++s) {
    const int dc = 16 * s + 8 * h;
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (qvalid[qs] && dc < HD)
      v = *reinterpret_cast<const u32x4*>(row_ptr<SEG>(a.q, a.q2, a.ldq, a.ldq2, a.n1, a.nq, b, q0 + 32 * qs + r) + head * HD + dc);
    // fold scale * log2(e) into Q once (re-rounded to the storage type): the MFMA then yields scores already in
    // log2 units, and the per-score FMA of the softmax disappears (VALU is this kernel's bound for d = 40)
    if (QSCALED) {
      v.x = T::pack2(T::to_f(v.x & 0xffff) * a.c, T::to_f(v.x >> 16) * a.c);
      v.y = T::pack2(T::to_f(v.y & 0xffff) * a.c, T::to_f(v.y >> 16) * a.c);
      v.z = T::pack2(T::to_f(v.z & 0xffff) * a.c, T::to_f(v.z >> 16) * a.c);
      v.w = T::pack2(T::to_f(v.w & 0xffff) * a.c, T::to_f(v.w >> 16) * a.c);
    }
    qf[qs][s] = *reinterpret_cast<typename T::v8*>(&v);
  }

  // ---- register-staged K/V tile loads: chunk e -> (matrix, row, 16-B chunk) ----
  // Loads are UNCONDITIONAL (clamped, always-valid addresses) so hipcc can count them (vmcnt(N)) and keep them in
  // flight across the MFMAs; rows past nk are zeroed when the registers are written to LDS.
  u32x4 stg[NLOAD];
  int ld_row[NLOAD], ld_dst[NLOAD];
  const unsigned short* ld_src[NLOAD];
  long ld_stride[NLOAD];
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    int e = tid + i * THREADS;
    const bool live = e < 2 * KV * CH;
    if (!live) e = 2 * KV * CH - 1;
    const int isv = e >= KV * CH;
    const int e2 = isv ? e - KV * CH : e;
    const int row = e2 / CH, ch = e2 - row * CH;
    ld_row[i] = row;
    ld_dst[i] = live ? (isv ? KV * KSTR + row * VSTR : row * KSTR) + ch * 16 : -1;
    ld_src[i] = (isv ? a.v + (long)b * a.nk * a.ldv : a.k + (long)b * a.nk * a.ldk) + head * HD + ch * 8;
    ld_stride[i] = isv ? a.ldv : a.ldk;
    if (SEG) { ld_stride[i] = isv; ld_src[i] = nullptr; ld_row[i] |= ch << 16; }    // segmented: resolved per tile
  }
  auto g_load = [&](int t) {
    const int k0 = t * KV;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      if (SEG) {
        const int krow = min(k0 + (ld_row[i] & 0xffff), a.nk - 1), ch = ld_row[i] >> 16;
        const unsigned short* rp = ld_stride[i] ? row_ptr<true>(a.v, a.v2, a.ldv, a.ldv2, a.n1, a.nk, b, krow)
                                                : row_ptr<true>(a.k, a.k2, a.ldk, a.ldk2, a.n1, a.nk, b, krow);
        stg[i] = *reinterpret_cast<const u32x4*>(rp + head * HD + ch * 8);
      } else {
        const int krow = min(k0 + ld_row[i], a.nk - 1);
        stg[i] = *reinterpret_cast<const u32x4*>(ld_src[i] + (long)krow * ld_stride[i]);
      }
    }
  };
  auto s_store = [&](int buf, int t) {
    const int k0 = t * KV;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      if (ld_dst[i] >= 0) {
        const u32x4 v = (k0 + (ld_row[i] & 0xffff) < a.nk) ? stg[i] : (u32x4){0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(smem + buf * STAGE + ld_dst[i]) = v;
      }
    }
  };

  // ---- LDS-DMA tile loads (DMA mode).  Piece p of a tile = chunks [64p, 64p+64) of the K image (p < KCH) or of the V
  // image; lane l fetches chunk 64p + l -> (row, chunk-in-row).  The per-lane source offset is FIXED for the whole
  // kernel: each tile moves the descriptor's base (scalar work) and shrinks its record count, so rows past nk and the
  // bank-pad chunks (offset 2^31) fail the range check and are written as zeros.
  constexpr int NPIECE = (KCH + VCH + 3) / 4;                  // pieces per wave per tile (4 waves)
  unsigned dma_off[NPIECE];
  if constexpr (DMA) {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int p = wid + 4 * i;                               // wave-uniform
      const bool isv = p >= KCH;
      const int e = (isv ? p - KCH : p) * 64 + lane;
      const int rowc = isv ? VCH : KCH;
      const int row = e / rowc, ch = e - row * rowc;
      dma_off[i] = ch < CH ? (unsigned)(((long)row * (isv ? a.ldv : a.ldk) + head * HD + ch * 8) * 2) : 0x80000000u;
    }
  }
  auto dma_issue = [&](int buf, int t) {
    const long k0 = (long)t * KV;
    const unsigned short *kb_, *vb_;
    long rk, rv;
    if (SEG && k0 >= a.n1) {                      // tile inside the second stream (host guarantees n1 % KV == 0, equal strides)
      const long kk = k0 - a.n1, n2 = a.nk - a.n1;
      kb_ = a.k2 + ((long)b * n2 + kk) * a.ldk2; vb_ = a.v2 + ((long)b * n2 + kk) * a.ldv2;
      rk = ((n2 - kk - 1) * a.ldk2 + a.heads * HD) * 2; rv = ((n2 - kk - 1) * a.ldv2 + a.heads * HD) * 2;
    } else {
      const long n1 = SEG ? a.n1 : a.nk;
      kb_ = a.k + ((long)b * n1 + k0) * a.ldk; vb_ = a.v + ((long)b * n1 + k0) * a.ldv;
      rk = ((n1 - k0 - 1) * a.ldk + a.heads * HD) * 2; rv = ((n1 - k0 - 1) * a.ldv + a.heads * HD) * 2;
    }
    const __amdgpu_buffer_rsrc_t rs_k = make_rsrc(kb_, (unsigned)(rk > 0 ? rk : 0));
    const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(vb_, (unsigned)(rv > 0 ? rv : 0));
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int p = wid + 4 * i;
      if (p < KCH)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lds_ptr_t)(smem + buf * STAGE + p * 1024), 16, dma_off[i], 0, 0, 0);
      else if (p < KCH + VCH)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_ptr_t)(smem + buf * STAGE + KV * KSTR + (p - KCH) * 1024), 16,
                                                 dma_off[i], 0, 0, 0);
    }
  };

  // Optimistic pass (bf16 only).  bf16 and fp32 share an exponent range, so p = 2^s needs no maximum at all as long as the
  // scores of a query stay inside the range -- and the per-tile maximum (16 v_max3 + a cross-half swap + two ballots per query
  // set and tile) is ~1/4 of the softmax's vector work in a loop that is bound by vector + matrix issue together.  The first
  // pass therefore runs WITHOUT it; the row sum (which the MFMA delivers anyway) tells afterwards whether that was sound:
  // 2^-64 <= sum_k p <= 2^64 bounds every p by 2^64 and puts the largest one above 2^-64 / nk, so nothing overflowed and no
  // term that matters underflowed (the same window as kGuard below).  Any query of the workgroup outside the window -- or a
  // NaN -- re-runs the block with the guarded loop (running maximum, re-centring).  SD / CLIP scores sit within +-30.
  // The two passes are two INSTANCES of the loop (compile-time flag): with one loop and a run-time flag the register
  // allocator carries both forms' live ranges at once (d = 40: 127 -> 166 VGPRs, one wave per SIMD fewer).
  SDN_ATS_DECL
  f32x16 o[QS][NDB];
  float l_tot[QS];
  auto run_pass = [&](auto safe_c) __attribute__((always_inline)) {
  constexpr bool safe = decltype(safe_c)::value;
  float m_run[QS], l_run[QS];                              // bf16: per-query exponent offset (0 = none); fp16: running max
  float m_hi[QS];                                          // bf16: running max of the scores (decides re-centring)
#pragma unroll
  for (int qs = 0; qs < QS; ++qs) {
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[qs][d][i] = 0.f;
    m_run[qs] = OFFSET_FREE ? 0.f : -1e30f; l_run[qs] = 0.f; m_hi[qs] = -1e30f;
    if constexpr (CENTRED) {                             // column HD of Q: 0 at the start of either pass (the guarded re-run wants it 0)
      u32x4 v = *reinterpret_cast<u32x4*>(&qf[qs][KQ - 1]);
      if (h == 1) v.x &= 0xffff0000u;
      qf[qs][KQ - 1] = *reinterpret_cast<typename T::v8*>(&v);
    }
  }

  // tr-read lane geometry
  const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, gcol = 16 * ((lane >> 4) & 1);

  const int ntiles = (a.nk + KV - 1) / KV;
  if constexpr (DMA) {
    dma_issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    g_load(0);
    s_store(0, 0);
  }
  __syncthreads();

  SDN_ATS_MARK(-1)
  SDN_ATS_PRO
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    const unsigned char* sK = smem + buf * STAGE;
    const unsigned char* sV = sK + KV * KSTR;
    if (t + 1 < ntiles) {                         // in flight during this tile's MFMAs / softmax (uniform branch)
      if constexpr (DMA) { if (!kLateDma) dma_issue(buf ^ 1, t + 1); }   // the other stage: last read before the previous barrier
      else g_load(t + 1);
    }

    // ---- S^T = K Q^T : two 32-key blocks ----
    f32x16 st[QS][2];
    // K fragments.  Left to itself hipcc reuses ONE 4-register fragment: read, wait, two MFMAs, read, wait ... -- every pair of
    // MFMAs behind a fresh LDS round trip.  With registers to spare (QS = 2: two waves per SIMD) all 2 x KQ fragments are
    // requested first, in one burst, and the scheduler is fenced from sinking them back.
    constexpr bool kPreloadK = QS == 2;
    typename T::v8 kfa[kPreloadK ? 2 : 1][kPreloadK ? KQ : 1];
    if constexpr (kPreloadK) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < KQ; ++s) {
          const unsigned char* kp = sK + (32 * kb + r) * KSTR + (16 * s + 8 * h) * 2;
          if (DMA && 16 * s + 16 > HD) { if (16 * s + 8 * h >= HD) kp = smem + ZOFF + (CENTRED ? 0 : 16); }   // zero padding of d (CENTRED: [1, 0 ...])
          kfa[kb][s] = *reinterpret_cast<const typename T::v8*>(kp);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(1);                // (MFMA clusters issue ahead of the SIMD's other waves: +3-4 % at d = 80, neutral at d = 40)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int qs = 0; qs < QS; ++qs)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[qs][kb][i] = 0.f;
#pragma unroll
      for (int s = 0; s < KQ; ++s) {
        typename T::v8 kf;
        if constexpr (kPreloadK) {
          kf = kfa[kb][s];
        } else {
          const unsigned char* kp = sK + (32 * kb + r) * KSTR + (16 * s + 8 * h) * 2;
          if (DMA && 16 * s + 16 > HD) { if (16 * s + 8 * h >= HD) kp = smem + ZOFF + (CENTRED ? 0 : 16); }   // zero padding of d (CENTRED: [1, 0 ...])
          kf = *reinterpret_cast<const typename T::v8*>(kp);
        }
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) st[qs][kb] = T::mfma32(kf, qf[qs][s], st[qs][kb]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (DMA && kLateDma) { if (t + 1 < ntiles) dma_issue(buf ^ 1, t + 1); }   // behind the QK^T MFMAs already in the pipe
    if constexpr (MASK) {
      const int k0 = t * KV;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = k0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
#pragma unroll
          for (int qs = 0; qs < QS; ++qs) {
            bool off = key >= a.nk || (a.causal && key > q0 + 32 * qs + r);
            if (!off && a.kmask) off = a.kmask[(long)b * a.nk + key] == 0;
            if (off) st[qs][kb][i] = -1e30f;
          }
        }
    } else if ((t + 1) * KV > a.nk) {             // ragged last tile (cross-attention: 77 keys)
      const int k0 = t * KV;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = k0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= a.nk) {
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) st[qs][kb][i] = -1e30f;
          }
        }
    }
    // ---- online softmax: the query is on the lane, its 32 keys of this tile are in st[0], st[1] ----
    SDN_ATS_MARK(0)                               // DMA issue + K reads + S MFMAs issued
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
    [[maybe_unused]] float mx = 0.f;
    bool need_mx = safe;                           // (a compile-time constant except in the centred fp16 pass: its first tile only)
    if constexpr ((CENTRED || CENTRED_SUB) && !safe) need_mx = t == 0;
    if (need_mx) {
    mx = fmaxf(st[qs][0][0], st[qs][1][0]);
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, st[qs][0][i]), st[qs][1][i]);
    {
      const unsigned u = __float_as_uint(mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // both halves of the same query
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    }
    if constexpr (OFFSET_FREE) {
    // Offset-free softmax: p = 2^(s - off) with a per-query offset that stays 0 while the tile maxima stay inside
    // +-kGuard (in log2 units; 2^+-64 is comfortably inside the fp32 / bf16 / fp16-scaled range used here), so the
    // common case needs NO subtraction and NO rescale of O -- softmax is invariant to the offset, the result is exact.
    // A tile whose maximum leaves the window re-centres that query's offset (O and the row sum are rescaled once).
    constexpr float kGuard = 64.f;
    m_hi[qs] = fmaxf(m_hi[qs], mx);                                    // running maximum of the scores seen so far
    const float drift = m_hi[qs] - m_run[qs];                          // m_run[qs] holds the current offset (0 until re-centred)
    if (safe && __builtin_amdgcn_ballot_w64(fabsf(drift) > kGuard) != 0) {       // wave-uniform, rare (safe: compile-time)
      // upward: the old sums shrink by 2^-(>64).  Downward can only happen on the first tile (m_hi[qs] never decreases),
      // when the sums are still zero -- the exponent is clamped so that 0 * alpha stays 0 instead of 0 * inf.
      const float m_new = fabsf(drift) > kGuard ? m_hi[qs] : m_run[qs];
      const float alpha = __builtin_amdgcn_exp2f(fminf(m_run[qs] - m_new, 126.f));
      if (!ONES) l_run[qs] *= alpha;
#pragma unroll
      for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[qs][d][i] *= alpha;
      m_run[qs] = m_new;
    }
    if (safe && __builtin_amdgcn_ballot_w64(m_run[qs] != 0.f) != 0) {                 // some query of this wave has an offset
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[qs][kb][i] = __builtin_amdgcn_exp2f(st[qs][kb][i] - m_run[qs]);
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[qs][kb][i] = __builtin_amdgcn_exp2f(st[qs][kb][i]);
    }
    if (!ONES) {
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) ps += st[qs][0][i] + st[qs][1][i];
      l_run[qs] += ps;
    }
    } else if constexpr (CENTRED || CENTRED_SUB) {
    if constexpr (!safe) {
      if (t == 0) {
        // the offset is the first tile's maximum ROUNDED to the storage type: tile 0 subtracts exactly what the MFMA will
        // subtract for the later tiles (CENTRED -- column HD: K reads 1.0, Q holds -off)
        const unsigned o16 = T::pack2(mx, 0.f) & 0xffffu;
        const float off = T::to_f(o16);
        m_run[qs] = off;
        if constexpr (CENTRED) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) st[qs][kb][i] -= off;
          u32x4 v = *reinterpret_cast<u32x4*>(&qf[qs][KQ - 1]);
          if (h == 1) v.x = (v.x & 0xffff0000u) | (o16 ^ 0x8000u);      // -off: the sign bit of the 16-bit pattern
          qf[qs][KQ - 1] = *reinterpret_cast<typename T::v8*>(&v);
        }
      }
      const float sub = CENTRED_SUB ? m_run[qs] : 0.f;                  // no free column: the subtraction per score, every tile
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[qs][kb][i] = __builtin_amdgcn_exp2f(CENTRED_SUB ? st[qs][kb][i] - sub : st[qs][kb][i]);
    } else {                                        // guarded re-run: running maximum on the pre-scaled scores (column HD of Q is 0)
      const float m_new = fmaxf(m_run[qs], mx);
      if (__builtin_amdgcn_ballot_w64(m_new != m_run[qs]) != 0) {            // wave-uniform: rescale only when needed
        const float alpha = __builtin_amdgcn_exp2f(m_run[qs] - m_new);
        if (!ONES) l_run[qs] *= alpha;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[qs][d][i] *= alpha;
        m_run[qs] = m_new;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[qs][kb][i] = __builtin_amdgcn_exp2f(st[qs][kb][i] - m_new);
    }
    if (!ONES) {
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) ps += st[qs][0][i] + st[qs][1][i];
      l_run[qs] += ps;
    }
    } else {                                          // fp16: classic running-max form (fp16 P needs p <= 1)
    const float m_new = fmaxf(m_run[qs], mx);
    const float mc = m_new * a.c;
    if (__builtin_amdgcn_ballot_w64(m_new != m_run[qs]) != 0) {              // wave-uniform: rescale only when needed
      const float alpha = __builtin_amdgcn_exp2f((m_run[qs] - m_new) * a.c);
      if (!ONES) l_run[qs] *= alpha;
#pragma unroll
      for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[qs][d][i] *= alpha;
      m_run[qs] = m_new;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) st[qs][kb][i] = __builtin_amdgcn_exp2f(fmaf(st[qs][kb][i], a.c, -mc));
    if (!ONES) {
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) ps += st[qs][0][i] + st[qs][1][i];
      l_run[qs] += ps;
    }
    }

    }  // query sets
    SDN_ATS_MARK(1)                               // max, exp (waits for the S MFMAs)
    // ---- O^T += V^T P^T ----
    // kAsmTr (d = 40, two query sets, LDS-DMA tiles): the V fragments are read by inline-asm ds_read_b64_tr_b16 behind hand-counted
    // lgkmcnt waits.  With the builtin, hipcc cannot prove that the next tile's LDS-DMA (in flight, other stage) does not alias
    // the reads and drains it -- s_waitcnt vmcnt(0) in the middle of every iteration.  Item k+1's two reads are issued before
    // item k is waited for: lgkmcnt(2) = "all but the two youngest LDS requests" (requests complete in order, so anything
    // else the compiler has in flight only makes the wait stricter); the waited registers are tied through the wait statement so
    // that no MFMA can be scheduled above it.
    constexpr bool kAsmTr = DMA && kAsmTrOn;
    __builtin_amdgcn_s_setprio(1);
    if constexpr (kAsmTr) {
      auto v_addr = [&](int item, int half) -> unsigned {
        const int kb = item / (2 * NDB), s2 = (item / NDB) & 1, d = item % NDB;
        const int keyb = 32 * kb + 16 * s2 + 4 * h + gq;
        const unsigned char* pa = sV + keyb * VSTR + (32 * d + gcol + 4 * gp) * 2;
        const unsigned char* pb = pa + 8 * VSTR;
        if (32 * d + 32 > HD) {
          const int col = 32 * d + gcol + 4 * gp;
          if (col >= HD) pa = pb = smem + ZOFF + (ONES && col == HD ? 0 : 8);
        }
        return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(half ? pb : pa);
      };
      s16x4 va[2][2];
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(va[0][0]) : "v"(v_addr(0, 0)));
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(va[0][1]) : "v"(v_addr(0, 1)));
#pragma unroll
      for (int item = 0; item < 4 * NDB; ++item) {
        const int kb = item / (2 * NDB), s2 = (item / NDB) & 1, d = item % NDB;
        if (item + 1 < 4 * NDB) {
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(va[(item + 1) & 1][0]) : "v"(v_addr(item + 1, 0)));
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(va[(item + 1) & 1][1]) : "v"(v_addr(item + 1, 1)));
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(va[item & 1][0]), "+v"(va[item & 1][1]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va[item & 1][0]), "+v"(va[item & 1][1]));
        }
        union { s16x4 hlf[2]; typename T::v8 full; } vf;
        vf.hlf[0] = va[item & 1][0]; vf.hlf[1] = va[item & 1][1];
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
          u32x4 pw;
          pw.x = T::pack2(st[qs][kb][8 * s2 + 0], st[qs][kb][8 * s2 + 1]);
          pw.y = T::pack2(st[qs][kb][8 * s2 + 2], st[qs][kb][8 * s2 + 3]);
          pw.z = T::pack2(st[qs][kb][8 * s2 + 4], st[qs][kb][8 * s2 + 5]);
          pw.w = T::pack2(st[qs][kb][8 * s2 + 6], st[qs][kb][8 * s2 + 7]);
          o[qs][d] = T::mfma32(vf.full, *reinterpret_cast<typename T::v8*>(&pw), o[qs][d]);
        }
      }
    } else {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        typename T::v8 pf[QS];
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
          u32x4 pw;
          pw.x = T::pack2(st[qs][kb][8 * s2 + 0], st[qs][kb][8 * s2 + 1]);
          pw.y = T::pack2(st[qs][kb][8 * s2 + 2], st[qs][kb][8 * s2 + 3]);
          pw.z = T::pack2(st[qs][kb][8 * s2 + 4], st[qs][kb][8 * s2 + 5]);
          pw.w = T::pack2(st[qs][kb][8 * s2 + 6], st[qs][kb][8 * s2 + 7]);
          pf[qs] = *reinterpret_cast<typename T::v8*>(&pw);
        }
        const int keyb = 32 * kb + 16 * s2 + 4 * h + gq;
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
          const unsigned char* pa = sV + keyb * VSTR + (32 * d + gcol + 4 * gp) * 2;
          const unsigned char* pb = pa + 8 * VSTR;
          if (DMA && 32 * d + 32 > HD) {               // columns past HD: the ones column, then zeros
            const int col = 32 * d + gcol + 4 * gp;
            if (col >= HD) pa = pb = smem + ZOFF + (ONES && col == HD ? 0 : 8);
          }
          union { s16x4 hlf[2]; typename T::v8 full; } vf;
          vf.hlf[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
          vf.hlf[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb));
#pragma unroll
          for (int qs = 0; qs < QS; ++qs) o[qs][d] = T::mfma32(vf.full, pf[qs], o[qs][d]);
        }
      }
    }

    __builtin_amdgcn_s_setprio(0);
    SDN_ATS_MARK(2)                               // pack, V tr reads, PV MFMAs issued
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // next tile landed; the barrier publishes it
    } else {
      if (t + 1 < ntiles) s_store(buf ^ 1, t + 1);  // the other stage: nobody reads it during this iteration
    }
    SDN_ATS_MARK(3)                               // wait for the next tile's DMA
    __syncthreads();
    SDN_ATS_MARK(4)                               // barrier
  }
  // ---- row sums; was the optimistic pass sound? ----
  bool bad = false;
#pragma unroll
  for (int qs = 0; qs < QS; ++qs) {
    if (ONES) {
      constexpr int row = HD % 32;                  // row of the last O^T block that accumulated sum_k p
      constexpr int ri = (row >> 3) * 4 + (row & 3);
      static_assert(((row >> 2) & 1) == 0, "ones row must live in lane half 0");
      l_tot[qs] = __shfl(o[qs][NDB - 1][ri], r, 64);      // lane r (half 0) holds it for query r
    } else {
      l_tot[qs] = l_run[qs] + __shfl_xor(l_run[qs], 32, 64);  // both halves hold partial sums of the same query
    }
    if constexpr (CENTRED || CENTRED_SUB) bad |= !(l_tot[qs] >= 0x1p-10f && l_tot[qs] <= 0x1p15f);   // every p <= sum <= 2^15 < fp16's maximum
    else bad |= !(l_tot[qs] >= 0x1p-64f && l_tot[qs] <= 0x1p64f);
  }
  return bad;
  };   // run_pass
  bool rerun = true;
  if constexpr (OFFSET_FREE || CENTRED || CENTRED_SUB) {
    if (a.nomax) {
      const bool bad = run_pass(std::false_type{});
      if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) *reinterpret_cast<volatile unsigned*>(smem + ZOFF + 32) = 1u;
      __syncthreads();
      rerun = *reinterpret_cast<volatile unsigned*>(smem + ZOFF + 32) != 0u;
    }
  }
  if (rerun) run_pass(std::true_type{});
  // ---- epilogue: normalise by the row sum ----
#pragma unroll
  for (int qs = 0; qs < QS; ++qs) {
  const float inv = 1.f / l_tot[qs];
  if (qvalid[qs]) {
    unsigned short* orow = const_cast<unsigned short*>(row_ptr<SEG>(a.out, a.out2, a.ldo, a.ldo2, a.n1, a.nq, b, q0 + 32 * qs + r)) + head * HD;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int tq = 0; tq < 4; ++tq) {
        const int dc = 32 * d + 8 * tq + 4 * h;
        if (dc < HD) {
          uint2 pk;
          pk.x = T::pack2(o[qs][d][4 * tq + 0] * inv, o[qs][d][4 * tq + 1] * inv);
          pk.y = T::pack2(o[qs][d][4 * tq + 2] * inv, o[qs][d][4 * tq + 3] * inv);
          *reinterpret_cast<uint2*>(orow + dc) = pk;
        }
      }
  }
  }
  SDN_ATS_MARK(6)                                 // epilogue: normalise + store
  SDN_ATS_FLUSH
#endif  // __HIP_DEVICE_COMPILE__
}

// ------------------------------------------------------------------------------------------------------------------
// bf16x3 self-attention on PRE-SPLIT operands (the precision mode's 64 x 64 / 32 x 32 levels).
//
// The bf16x3 plan's first attention kernel (k_attention_x3, sdn_f32.hip) takes f32 Q / K / V and splits every K / V tile
// into bf16 hi + lo on its way into LDS -- per workgroup, i.e. 32 times per tensor at N = 4096, through registers and
// 2-byte LDS scatter writes (151-184 TFLOP/s).  Here the qkv projection writes its output as hi | lo PAIR rows
// (sdn_gemm_desc.x3_out = 4: row = [hi(3C) | lo(3C)], the same 4 bytes per element as f32), and this kernel is k_attn's
// LDS-DMA design run three times per product:
//   S^T  = Kl Qh^T + Kh Ql^T + Kh Qh^T          (32x32x16 bf16 MFMAs, small terms first; K planes by LDS-DMA, Q in registers)
//   O^T += Vl^T Ph^T + Vh^T Pl^T + Vh^T Ph^T    (P split in registers: ph = bf16(p), pl = bf16(p - ph); V^T by tr reads)
// Scores stay unscaled (scaling a pre-split hi / lo pair would re-round it): p = 2^(s c - m c) with c = scale log2 e.
// The row sum rides in the ones column of the V-hi padding (sum ph + sum pl).  Running maximum, wave-uniform rescale.
struct AttnPArgs {
  const unsigned short* q; const unsigned short* k; const unsigned short* v;   // hi planes; the lo plane of a row lies `lo` elements on
  void* out;
  int nq, nk, ldq, ldk, ldv, ldo, lo_q, lo_kv;                                 // lo plane of a Q row / of a K or V row: that many elements on
  float c;
  int heads, nqb, npairs, triple;
};

// NW = waves per workgroup (4 or 8: d = 80 holds 94 KB of K / V stages, one workgroup per CU -- eight waves share them)
// KVT = keys per tile (64; 32 at d = 160, whose four plane images of 64 keys would not leave room for two stages)
template <int HD, int QS, int NW, int KVT>
__global__ void __launch_bounds__(64 * NW)
k_attn_x3p(const AttnPArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef SdnBF16 T;
  constexpr int KQ = (HD + 15) / 16, NDB = (HD + 31) / 32, CH = HD / 8;
  constexpr bool ONES = HD % 32 != 0;                           // row sums from the ones column of the padding (else summed on the vector side)
  constexpr int KCH = HD == 40 ? 5 : CH + 1, VCH = HD == 40 ? 5 : (HD == 160 ? 20 : 12);
  constexpr int KSTR = KCH * 16, VSTR = VCH * 16, NKB = KVT / 32;
  constexpr int KP = (KVT * KCH + 63) / 64, VP = (KVT * VCH + 63) / 64;      // 1-KiB DMA pieces per K / V plane image
  constexpr int OFF_KL = KP * 1024, OFF_VH = 2 * KP * 1024, OFF_VL = OFF_VH + VP * 1024;
  constexpr int STAGE = 2 * (KP + VP) * 1024;
  constexpr int ZOFF = 2 * STAGE;
  constexpr int NP = 2 * (KP + VP), NPIECE = (NP + NW - 1) / NW;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE + 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  int pair, qblk;
  {
    const int nqb = a.nqb, id = blockIdx.x;
    if ((a.npairs & 7) == 0) { const int x = id & 7, j = id >> 3; pair = (j / nqb) * 8 + x; qblk = j - (j / nqb) * nqb; }
    else { pair = id / nqb; qblk = id - pair * nqb; }
  }
  const int head = pair % a.heads, b = pair / a.heads;
  const int q0 = qblk * (32 * NW * QS) + wid * (32 * QS);
  bool qvalid[QS];
#pragma unroll
  for (int qs = 0; qs < QS; ++qs) qvalid[qs] = q0 + 32 * qs + r < a.nq;
  if (tid < 8) *reinterpret_cast<unsigned*>(smem + ZOFF + tid * 4) = tid == 0 ? (T::pack2(1.0f, 0.0f) & 0xffffu) : 0u;

  // Q fragments, hi and lo (B operand: lane (r, h) holds Q[q0 + r][16 s + 8 h .. + 8))
  typename T::v8 qh[QS][KQ], ql[QS][KQ];
#pragma unroll
  for (int qs = 0; qs < QS; ++qs)
#pragma unroll
    for (int s = 0; s < KQ; ++s) {
      const int dc = 16 * s + 8 * h;
      u32x4 vh = (u32x4){0u, 0u, 0u, 0u}, vl = vh;
      if (qvalid[qs] && dc < HD) {
        const unsigned short* qp = a.q + ((long)b * a.nq + q0 + 32 * qs + r) * a.ldq + head * HD + dc;
        vh = *reinterpret_cast<const u32x4*>(qp); vl = *reinterpret_cast<const u32x4*>(qp + a.lo_q);
      }
      qh[qs][s] = *reinterpret_cast<typename T::v8*>(&vh); ql[qs][s] = *reinterpret_cast<typename T::v8*>(&vl);
    }

  // tile images in a stage: [K hi | K lo | V hi | V lo]; piece p = 64 chunks of one image (as k_attn's DMA mode)
  unsigned dma_off[NPIECE];
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) {
    const int p = wid + NW * i;                                   // wave-uniform
    const bool isv = p >= 2 * KP;
    const int pi = isv ? (p - 2 * KP) % VP : p % KP;             // piece inside its image
    const int e = pi * 64 + lane;
    const int rowc = isv ? VCH : KCH;
    const int row = e / rowc, ch = e - row * rowc;
    dma_off[i] = (ch < CH && row < KVT) ? (unsigned)(((long)row * (isv ? a.ldv : a.ldk) + head * HD + ch * 8) * 2) : 0x80000000u;
  }
  auto dma_issue = [&](int buf, int t) {
    const long k0 = (long)t * KVT;
    const unsigned short* kb_ = a.k + ((long)b * a.nk + k0) * a.ldk;
    const unsigned short* vb_ = a.v + ((long)b * a.nk + k0) * a.ldv;
    const long rk = ((a.nk - k0 - 1) * a.ldk + a.heads * HD) * 2, rv = ((a.nk - k0 - 1) * a.ldv + a.heads * HD) * 2;
    const __amdgpu_buffer_rsrc_t rs_kh = make_rsrc(kb_, (unsigned)(rk > 0 ? rk : 0));
    const __amdgpu_buffer_rsrc_t rs_kl = make_rsrc(kb_ + a.lo_kv, (unsigned)(rk > 0 ? rk : 0));
    const __amdgpu_buffer_rsrc_t rs_vh = make_rsrc(vb_, (unsigned)(rv > 0 ? rv : 0));
    const __amdgpu_buffer_rsrc_t rs_vl = make_rsrc(vb_ + a.lo_kv, (unsigned)(rv > 0 ? rv : 0));
    unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      const int p = wid + NW * i;
      if (p < KP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kh, (lds_ptr_t)(st + p * 1024), 16, dma_off[i], 0, 0, 0);
      else if (p < 2 * KP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kl, (lds_ptr_t)(st + OFF_KL + (p - KP) * 1024), 16, dma_off[i], 0, 0, 0);
      else if (p < 2 * KP + VP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_vh, (lds_ptr_t)(st + OFF_VH + (p - 2 * KP) * 1024), 16, dma_off[i], 0, 0, 0);
      else if (p < NP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_vl, (lds_ptr_t)(st + OFF_VL + (p - 2 * KP - VP) * 1024), 16, dma_off[i], 0, 0, 0);
    }
  };

  f32x16 o[QS][NDB];
  float m_run[QS];
  [[maybe_unused]] float l_run[QS];
#pragma unroll
  for (int qs = 0; qs < QS; ++qs) {
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[qs][d][i] = 0.f;
    m_run[qs] = -1e30f; l_run[qs] = 0.f;
  }
  const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, gcol = 16 * ((lane >> 4) & 1);
  const int ntiles = (a.nk + KVT - 1) / KVT;
  dma_issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    const unsigned char* sKh = smem + buf * STAGE;
    const unsigned char* sVh = sKh + OFF_VH;
    // ---- S^T = K Q^T, three products per k-step ----
    f32x16 st[QS][NKB];
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int qs = 0; qs < QS; ++qs)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[qs][kb][i] = 0.f;
#pragma unroll
      for (int s = 0; s < KQ; ++s) {
        const unsigned char* kp = sKh + (32 * kb + r) * KSTR + (16 * s + 8 * h) * 2;
        const unsigned char* kq = kp + OFF_KL;
        if (16 * s + 16 > HD) { if (16 * s + 8 * h >= HD) kp = kq = smem + ZOFF + 16; }   // zero padding of d
        const typename T::v8 kfh = *reinterpret_cast<const typename T::v8*>(kp), kfl = *reinterpret_cast<const typename T::v8*>(kq);
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) st[qs][kb] = T::mfma32(kfl, qh[qs][s], st[qs][kb]);
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) st[qs][kb] = T::mfma32(kfh, ql[qs][s], st[qs][kb]);
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) st[qs][kb] = T::mfma32(kfh, qh[qs][s], st[qs][kb]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (t + 1 < ntiles) dma_issue(buf ^ 1, t + 1);            // behind the QK^T MFMAs already in the pipe; lands under softmax + PV
    if ((t + 1) * KVT > a.nk) {                               // ragged last tile
      const int k0 = t * KVT;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = k0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= a.nk) {
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) st[qs][kb][i] = -1e30f;
          }
        }
    }
    // ---- online softmax (running maximum, wave-uniform rescale) ----
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
      float mx = st[qs][0][0];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[qs][kb][i]);
      {
        const unsigned u = __float_as_uint(mx);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      const float m_new = fmaxf(m_run[qs], mx);
      const float mc = m_new * a.c;
      if (__builtin_amdgcn_ballot_w64(m_new != m_run[qs]) != 0) {
        const float alpha = __builtin_amdgcn_exp2f((m_run[qs] - m_new) * a.c);
        if (!ONES) l_run[qs] *= alpha;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[qs][d][i] *= alpha;
        m_run[qs] = m_new;
      }
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[qs][kb][i] = __builtin_amdgcn_exp2f(fmaf(st[qs][kb][i], a.c, -mc));
      if (!ONES) {
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) ps += st[qs][kb][i];
        l_run[qs] += ps;
      }
    }
    // ---- O^T += V^T P^T, three products per (16-key step, 32-dim block) ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        typename T::v8 ph[QS], pl[QS];
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
          u32x4 wh, wl;
          const float* pv = reinterpret_cast<const float*>(&st[qs][kb]) + 8 * s2;
          unsigned hh[4], ll[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hh[e] = T::pack2(pv[2 * e], pv[2 * e + 1]);
            ll[e] = T::pack2(pv[2 * e] - T::to_f(hh[e] & 0xffff), pv[2 * e + 1] - T::to_f(hh[e] >> 16));
          }
          wh.x = hh[0]; wh.y = hh[1]; wh.z = hh[2]; wh.w = hh[3];
          wl.x = ll[0]; wl.y = ll[1]; wl.z = ll[2]; wl.w = ll[3];
          ph[qs] = *reinterpret_cast<typename T::v8*>(&wh); pl[qs] = *reinterpret_cast<typename T::v8*>(&wl);
        }
        const int keyb = 32 * kb + 16 * s2 + 4 * h + gq;
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
          const unsigned char* pa = sVh + keyb * VSTR + (32 * d + gcol + 4 * gp) * 2;
          const unsigned char* pb = pa + 8 * VSTR;
          const unsigned char* la = pa + (OFF_VL - OFF_VH);
          const unsigned char* lb = pb + (OFF_VL - OFF_VH);
          if (32 * d + 32 > HD) {                               // columns past HD: V hi = the ones column, then zeros; V lo = zeros
            const int col = 32 * d + gcol + 4 * gp;
            if (col >= HD) { pa = pb = smem + ZOFF + ((ONES && col == HD) ? 0 : 8); la = lb = smem + ZOFF + 8; }
          }
          union { s16x4 hlf[2]; typename T::v8 full; } vfh, vfl;
          vfh.hlf[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
          vfh.hlf[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb));
          vfl.hlf[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(la));
          vfl.hlf[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lb));
#pragma unroll
          for (int qs = 0; qs < QS; ++qs) o[qs][d] = T::mfma32(vfl.full, ph[qs], o[qs][d]);
#pragma unroll
          for (int qs = 0; qs < QS; ++qs) o[qs][d] = T::mfma32(vfh.full, pl[qs], o[qs][d]);
#pragma unroll
          for (int qs = 0; qs < QS; ++qs) o[qs][d] = T::mfma32(vfh.full, ph[qs], o[qs][d]);
        }
      }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // ---- epilogue: normalise by the row sum, write f32 rows or [hi | lo | hi] triples ----
#pragma unroll
  for (int qs = 0; qs < QS; ++qs) {
    float l_tot;
    if constexpr (ONES) {
      constexpr int row = HD % 32, ri = (row >> 3) * 4 + (row & 3);
      static_assert(((row >> 2) & 1) == 0, "ones row must live in lane half 0");
      l_tot = __shfl(o[qs][NDB - 1][ri], r, 64);
    } else {
      l_tot = l_run[qs] + __shfl_xor(l_run[qs], 32, 64);      // both halves hold partial sums of the same query
    }
    const float inv = 1.f / l_tot;
    if (!qvalid[qs]) continue;
    const long qrow = (long)b * a.nq + q0 + 32 * qs + r;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int tq = 0; tq < 4; ++tq) {
        const int dc = 32 * d + 8 * tq + 4 * h;
        if (dc < HD) {
          const float v0 = o[qs][d][4 * tq + 0] * inv, v1 = o[qs][d][4 * tq + 1] * inv, v2 = o[qs][d][4 * tq + 2] * inv, v3 = o[qs][d][4 * tq + 3] * inv;
          if (a.triple) {
            unsigned short* o3 = reinterpret_cast<unsigned short*>(a.out) + qrow * 3 * a.ldo + head * HD + dc;
            uint2 hi, lo;
            hi.x = T::pack2(v0, v1); hi.y = T::pack2(v2, v3);
            lo.x = T::pack2(v0 - T::to_f(hi.x & 0xffff), v1 - T::to_f(hi.x >> 16));
            lo.y = T::pack2(v2 - T::to_f(hi.y & 0xffff), v3 - T::to_f(hi.y >> 16));
            *reinterpret_cast<uint2*>(o3) = hi; *reinterpret_cast<uint2*>(o3 + a.ldo) = lo; *reinterpret_cast<uint2*>(o3 + 2 * a.ldo) = hi;
          } else {
            *reinterpret_cast<sdn_f32x4*>(reinterpret_cast<float*>(a.out) + qrow * a.ldo + head * HD + dc) = (sdn_f32x4){v0, v1, v2, v3};
          }
        }
      }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

static int g_attn_qs2 = 1;           // debug A/B switch (sdn_debug_set_attn_qs2)
static int g_attn_nomax = 1;         // debug A/B switch (sdn_debug_set_attn_nomax): 0 = guarded pass only

template <typename T, int HD>
int launch(const AttnArgs& a, int batch, int heads, hipStream_t st) {
  if (a.causal || a.kmask) {
    if constexpr (HD == 64) {
      if (a.q2) return SDN_E_INVALID;
      hipLaunchKernelGGL((k_attn<T, HD, false, true>), dim3(a.nqb * a.npairs), dim3(THREADS), 0, st, a);
      return sdn_launch_status();
    } else {
      return SDN_E_INVALID;                     // masked attention is instantiated for d = 64 (CLIP) only
    }
  }
  if (a.q2) {
    if constexpr (HD == 64) {                     // MMDiT head dim: LDS-DMA variant when no tile straddles the two streams
      if (a.n1 % KV == 0 && a.ldk == a.ldk2 && a.ldv == a.ldv2) {
        // (two query sets per wave were measured here too -- MMDiT joint attention, d = 64, fp16: 13.1 vs 12.3 ms per forward at
        //  batch 32: the halved LDS traffic does not pay for the halved occupancy when no padding column inflates the MFMA count)
        hipLaunchKernelGGL((k_attn<T, HD, true, false, true>), dim3(a.nqb * a.npairs), dim3(THREADS), 0, st, a);
        return sdn_launch_status();
      }
    }
    hipLaunchKernelGGL((k_attn<T, HD, true>), dim3(a.nqb * a.npairs), dim3(THREADS), 0, st, a);
  } else {
    if constexpr (HD == 40) {
      // long key sets at d = 40 (the 64 x 64 self-attention, 15 % of the forward): two query sets per wave (see QS)
      if (a.nk > 2 * KV && a.nq >= 2 * QB && g_attn_qs2) {
        AttnArgs a2 = a;
        a2.nqb = (a.nq + 2 * QB - 1) / (2 * QB);
        hipLaunchKernelGGL((k_attn<T, HD, false, false, true, 2>), dim3(a2.nqb * a2.npairs), dim3(THREADS), 0, st, a2);
        return sdn_launch_status();
      }
    }
    hipLaunchKernelGGL((k_attn<T, HD, false>), dim3(a.nqb * a.npairs), dim3(THREADS), 0, st, a);
  }
  return sdn_launch_status();
}

static int g_attn_head_inner = 1;    // debug A/B switch (sdn_debug_set_attn_head_inner)

template <typename T>
int run(const void* q, const void* k, const void* v, void* out, int32_t batch, int32_t heads, int32_t nq, int32_t nk,
        int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, float scale, void* stream,
        const sdn_attn_segment2* s2 = nullptr, int causal = 0, const int* kmask = nullptr) {
  if (!q || !k || !v || !out || batch < 0 || heads <= 0 || nq <= 0 || nk <= 0) return SDN_E_INVALID;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) || nk > 65535) return SDN_E_INVALID;
  if (ldq < heads * head_dim || ldk < heads * head_dim || ldv < heads * head_dim || ldo < heads * head_dim)
    return SDN_E_INVALID;
  auto al = [](const void* p, int n) { return (reinterpret_cast<uintptr_t>(p) & (n - 1)) == 0; };
  if (!al(q, 16) || !al(k, 16) || !al(v, 16) || !al(out, 8)) return SDN_E_INVALID;
  if (batch == 0) return SDN_OK;
  AttnArgs a{(const unsigned short*)q, (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)out,
             nq, nk, ldq, ldk, ldv, ldo, scale * 1.4426950408889634f, heads, (nq + QB - 1) / QB, batch * heads,
             nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0, causal, kmask, g_attn_nomax};
  if (!s2 && nk <= 2 * KV && (batch & 7) == 0 && g_attn_head_inner) a.head_inner = 1;
  if (causal && nq != nk) return SDN_E_INVALID;
  if (s2) {                                                  // joint attention over two token streams (nq == nk)
    if (!s2->q2 || !s2->k2 || !s2->v2 || !s2->out2 || s2->n1 <= 0 || s2->n1 >= nq || nq != nk) return SDN_E_INVALID;
    if ((s2->ldq2 & 7) || (s2->ldk2 & 7) || (s2->ldv2 & 7) || (s2->ldo2 & 3)) return SDN_E_INVALID;
    if (!al(s2->q2, 16) || !al(s2->k2, 16) || !al(s2->v2, 16) || !al(s2->out2, 8)) return SDN_E_INVALID;
    a.q2 = (const unsigned short*)s2->q2; a.k2 = (const unsigned short*)s2->k2; a.v2 = (const unsigned short*)s2->v2;
    a.out2 = (unsigned short*)s2->out2; a.n1 = s2->n1;
    a.ldq2 = s2->ldq2; a.ldk2 = s2->ldk2; a.ldv2 = s2->ldv2; a.ldo2 = s2->ldo2;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (head_dim) {
    case 40: return launch<T, 40>(a, batch, heads, st);
    case 64: return launch<T, 64>(a, batch, heads, st);
    case 80: return launch<T, 80>(a, batch, heads, st);
    case 160: return launch<T, 160>(a, batch, heads, st);
    default: return SDN_E_INVALID;
  }
}

}  // namespace sdn_attn_detail

#ifdef SDN_ATTN_STAMPS
extern "C" int sdn_debug_set_attn_stamps(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(sdn_attn_detail::g_attn_stamps), &p, sizeof(p));
}
#endif
extern "C" void sdn_debug_set_attn_qs2(int on) { sdn_attn_detail::g_attn_qs2 = on; }
extern "C" void sdn_debug_set_attn_nomax(int on) { sdn_attn_detail::g_attn_nomax = on; }
extern "C" int sdn_attention_bf16(const void* q, const void* k, const void* v, void* out, int32_t batch,
                                  int32_t heads, int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq,
                                  int32_t ldk, int32_t ldv, int32_t ldo, float scale, void* stream) {
  return sdn_attn_detail::run<SdnBF16>(q, k, v, out, batch, heads, nq, nk, head_dim, ldq, ldk, ldv, ldo, scale, stream);
}
extern "C" int sdn_attention_f16(const void* q, const void* k, const void* v, void* out, int32_t batch,
                                 int32_t heads, int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq,
                                 int32_t ldk, int32_t ldv, int32_t ldo, float scale, void* stream) {
  return sdn_attn_detail::run<SdnF16>(q, k, v, out, batch, heads, nq, nk, head_dim, ldq, ldk, ldv, ldo, scale, stream);
}
extern "C" int sdn_joint_attention(int32_t dtype, const void* q, const void* k, const void* v, void* out,
                                   const sdn_attn_segment2* seg2, int32_t batch, int32_t heads, int32_t n_total,
                                   int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, float scale,
                                   void* stream) {
  if (!seg2) return SDN_E_INVALID;
  if (dtype == 1)
    return sdn_attn_detail::run<SdnF16>(q, k, v, out, batch, heads, n_total, n_total, head_dim, ldq, ldk, ldv, ldo, scale,
                                        stream, seg2);
  return sdn_attn_detail::run<SdnBF16>(q, k, v, out, batch, heads, n_total, n_total, head_dim, ldq, ldk, ldv, ldo, scale,
                                       stream, seg2);
}

extern "C" int sdn_masked_attention(int32_t dtype, const void* q, const void* k, const void* v, void* out,
                                    const int32_t* key_mask, int32_t causal, int32_t batch, int32_t heads, int32_t n,
                                    int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, float scale,
                                    void* stream) {
  if (!causal && !key_mask) return SDN_E_INVALID;
  if (dtype == 1)
    return sdn_attn_detail::run<SdnF16>(q, k, v, out, batch, heads, n, n, head_dim, ldq, ldk, ldv, ldo, scale, stream, nullptr,
                                        causal ? 1 : 0, key_mask);
  return sdn_attn_detail::run<SdnBF16>(q, k, v, out, batch, heads, n, n, head_dim, ldq, ldk, ldv, ldo, scale, stream, nullptr,
                                       causal ? 1 : 0, key_mask);
}

extern "C" void sdn_debug_set_attn_head_inner(int on) { sdn_attn_detail::g_attn_head_inner = on; }

// bf16x3 self-attention on hi | lo pair rows (sdn.h).  q / k / v point at the hi planes; the lo plane of every row lies `lo_offset`
// elements on (the qkv projection's pair output: ld = 6 C, lo_offset = 3 C).  triple = 0: out f32 [B, nq, ldo]; 1: [hi | lo | hi] rows.
extern "C" int sdn_attention_x3_pairs(const void* q, const void* k, const void* v, int32_t lo_offset_q, int32_t lo_offset_kv, void* out, int32_t batch,
                                      int32_t heads, int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq, int32_t ldk, int32_t ldv,
                                      int32_t ldo, float scale, int32_t triple, void* stream) {
  using namespace sdn_attn_detail;
  if (!q || !k || !v || !out || batch < 0 || heads <= 0 || nq <= 0 || nk <= 0 || lo_offset_q <= 0 || lo_offset_kv <= 0) return SDN_E_INVALID;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (lo_offset_q & 7) || (lo_offset_kv & 7) || (ldo & 3) || nk > 65535) return SDN_E_INVALID;
  if (lo_offset_q < heads * head_dim || lo_offset_kv < heads * head_dim || ldq < lo_offset_q + heads * head_dim ||
      ldk < lo_offset_kv + heads * head_dim || ldv < lo_offset_kv + heads * head_dim || ldo < heads * head_dim)
    return SDN_E_INVALID;
  auto al = [](const void* p, int n) { return (reinterpret_cast<uintptr_t>(p) & (n - 1)) == 0; };
  if (!al(q, 16) || !al(k, 16) || !al(v, 16) || !al(out, 16)) return SDN_E_INVALID;
  if (head_dim != 40 && head_dim != 80 && head_dim != 160) return SDN_E_INVALID;
  if ((long)nk * ldk * 2 >= (1L << 31) || (long)nk * ldv * 2 >= (1L << 31)) return SDN_E_INVALID;     // 31-bit DMA offsets per sample
  if (batch == 0) return SDN_OK;
  static const int qs_env = getenv("SDN_X3P_QS") ? atoi(getenv("SDN_X3P_QS")) : 0;      /* tuning knobs (tools/profile_x3.py) */
  static const int nw_env = getenv("SDN_X3P_NW") ? atoi(getenv("SDN_X3P_NW")) : 0;
  int qs = (head_dim == 40 && nq >= 2 * QB) ? 2 : 1;
  int nw = (head_dim == 80 || head_dim == 160) && nq >= 2 * QB ? 8 : 4;
  if (head_dim == 40 && qs_env) qs = qs_env == 2 ? 2 : 1;
  if (nw_env) nw = nw_env == 8 ? 8 : 4;
  if (head_dim != 40) qs = 1;
  if (head_dim == 40 && qs == 2) nw = 4;
  const int qper = 32 * nw * qs;
  AttnPArgs a{(const unsigned short*)q, (const unsigned short*)k, (const unsigned short*)v, out, nq, nk, ldq, ldk, ldv, ldo, lo_offset_q, lo_offset_kv,
              scale * 1.4426950408889634f, heads, (nq + qper - 1) / qper, batch * heads, triple ? 1 : 0};
  const unsigned grid = (unsigned)(a.nqb * a.npairs);
  hipStream_t st = (hipStream_t)stream;
  if (head_dim == 40) {
    if (qs == 2) hipLaunchKernelGGL((k_attn_x3p<40, 2, 4, 64>), dim3(grid), dim3(256), 0, st, a);
    else if (nw == 8) hipLaunchKernelGGL((k_attn_x3p<40, 1, 8, 64>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_attn_x3p<40, 1, 4, 64>), dim3(grid), dim3(256), 0, st, a);
  } else if (head_dim == 80) {
    if (nw == 8) hipLaunchKernelGGL((k_attn_x3p<80, 1, 8, 64>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_attn_x3p<80, 1, 4, 64>), dim3(grid), dim3(256), 0, st, a);
  } else {
    if (nw == 8) hipLaunchKernelGGL((k_attn_x3p<160, 1, 8, 32>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_attn_x3p<160, 1, 4, 32>), dim3(grid), dim3(256), 0, st, a);
  }
  return sdn_launch_status();
}
