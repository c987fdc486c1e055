// Fused attention forward (row U6): softmax(Q K^T * scale) V, flash-style, bf16 in / fp32 accumulate,
// v_mfma_f32_32x32x16_bf16.  Head dims 40/80/160 (SD-v1.4: 8 heads at C=320/640/1280) and 64 (MMDiT).
//
// Orientation (CDNA4-specific, see the guide's "accumulator tile as the next MFMA's operand"):
//   S^T = K . Q^T      A = K rows from LDS (ds_read_b128), B = Q rows held in registers for the whole kernel
//                      -> a lane owns ONE query column (lane & 31) and 16 keys per 32-key block in registers,
//                         so the softmax max/sum are per-lane loops + one cross-half shuffle;
//   O^T += V^T . P^T   B = the S^T accumulator itself, converted to bf16 in registers (no LDS round trip),
//                      A = V^T read from the row-major V tile with ds_read_b64_tr_b16 (hardware transpose).
// The running max / sum and the O^T rescale are per-lane scalars because the query sits on the lane.
// Head dim is zero-padded inside LDS/registers only (40 -> 48 for QK^T, 64 for PV); HBM tensors stay packed
// [B, N, heads*d], i.e. the NHWC token matrix the projections write.
#include "sdn_common.h"
#include "sdn_ops.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int THREADS = 256;
constexpr int KV = 64;       // keys per tile
constexpr int QB = 128;      // queries per workgroup (32 per wave)

struct AttnArgs {
  const unsigned short* q; const unsigned short* k; const unsigned short* v; unsigned short* out;
  int nq, nk, ldq, ldk, ldv, ldo;
  float c;                   // scale * log2(e)
};

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 p = {(__bf16)lo, (__bf16)hi};
  return *reinterpret_cast<unsigned*>(&p);
}

template <int HD>
__global__ void __launch_bounds__(THREADS)
k_attn(const AttnArgs a) {
  constexpr int KQ = (HD + 15) / 16;          // 16-deep k-steps of Q K^T
  constexpr int NDB = (HD + 31) / 32;         // 32-row blocks of O^T
  constexpr int KSTR = KQ * 32 + 16;          // K tile row stride in bytes (+16 B pad against bank conflicts)
  constexpr int VSTR = NDB * 64 + 16;         // V tile row stride in bytes (multiple of 8 for the tr read)
  __shared__ __attribute__((aligned(16))) unsigned char smem[KV * KSTR + KV * VSTR];
  unsigned char* sK = smem;
  unsigned char* sV = smem + KV * KSTR;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * QB + wid * 32;
  const bool qvalid = q0 < a.nq;

  // ---- Q fragments (B operand: lane (r,h) holds Q[q0+r][16s + 8h .. +8)) ----
  bf16x8 qf[KQ];
#pragma unroll
  for (int s = 0; s < KQ; ++s) {
    const int dc = 16 * s + 8 * h;
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    if (qvalid && dc < HD) v = *reinterpret_cast<const u32x4*>(a.q + ((long)b * a.nq + q0 + r) * a.ldq + head * HD + dc);
    qf[s] = *reinterpret_cast<bf16x8*>(&v);
  }

  f32x16 o[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // tr-read lane geometry
  const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, gcol = 16 * ((lane >> 4) & 1);

  const int ntiles = (a.nk + KV - 1) / KV;
  for (int t = 0; t < ntiles; ++t) {
    const int k0 = t * KV;
    __syncthreads();
    // ---- cooperative K / V tile load, zero-filled outside [0,nk) x [0,HD) ----
    for (int e = tid; e < KV * KQ * 2; e += THREADS) {
      const int row = e / (KQ * 2), ch = e - row * (KQ * 2);
      u32x4 v = (u32x4){0u, 0u, 0u, 0u};
      if (k0 + row < a.nk && ch * 8 < HD)
        v = *reinterpret_cast<const u32x4*>(a.k + ((long)b * a.nk + k0 + row) * a.ldk + head * HD + ch * 8);
      *reinterpret_cast<u32x4*>(sK + row * KSTR + ch * 16) = v;
    }
    for (int e = tid; e < KV * NDB * 4; e += THREADS) {
      const int row = e / (NDB * 4), ch = e - row * (NDB * 4);
      u32x4 v = (u32x4){0u, 0u, 0u, 0u};
      if (k0 + row < a.nk && ch * 8 < HD)
        v = *reinterpret_cast<const u32x4*>(a.v + ((long)b * a.nk + k0 + row) * a.ldv + head * HD + ch * 8);
      *reinterpret_cast<u32x4*>(sV + row * VSTR + ch * 16) = v;
    }
    __syncthreads();

    // ---- S^T = K Q^T : two 32-key blocks ----
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) st[kb][i] = 0.f;
#pragma unroll
      for (int s = 0; s < KQ; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (32 * kb + r) * KSTR + (16 * s + 8 * h) * 2);
        st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[kb], 0, 0, 0);
      }
    }
    if (k0 + KV > a.nk) {                        // ragged last tile (cross-attention: 77 keys)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = k0 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= a.nk) st[kb][i] = -1e30f;
        }
    }
    // ---- online softmax: the query is on the lane, its 32 keys of this tile are in st[0], st[1] ----
    float mx = st[0][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, st[0][i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[1][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f((m_run - m_new) * a.c);
    m_run = m_new;
    const float mc = m_new * a.c;
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = exp2f(fmaf(st[kb][i], a.c, -mc));
        st[kb][i] = p;
        ps += p;
      }
    l_run = l_run * alpha + ps;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[d][i] *= alpha;

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 pw;
        pw.x = pack2(st[kb][8 * s2 + 0], st[kb][8 * s2 + 1]);
        pw.y = pack2(st[kb][8 * s2 + 2], st[kb][8 * s2 + 3]);
        pw.z = pack2(st[kb][8 * s2 + 4], st[kb][8 * s2 + 5]);
        pw.w = pack2(st[kb][8 * s2 + 6], st[kb][8 * s2 + 7]);
        const bf16x8 pf = *reinterpret_cast<bf16x8*>(&pw);
        const int keyb = 32 * kb + 16 * s2 + 4 * h + gq;
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
          const unsigned char* pa = sV + keyb * VSTR + (32 * d + gcol + 4 * gp) * 2;
          union { s16x4 hlf[2]; bf16x8 full; } vf;
          vf.hlf[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
          vf.hlf[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 8 * VSTR));
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.full, pf, o[d], 0, 0, 0);
        }
      }
  }

  // ---- epilogue: normalise by the row sum (both lane halves hold partial sums of the same query) ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (qvalid) {
    unsigned short* orow = a.out + ((long)b * a.nq + q0 + r) * a.ldo + head * HD;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int tq = 0; tq < 4; ++tq) {
        const int dc = 32 * d + 8 * tq + 4 * h;
        if (dc < HD) {
          uint2 pk;
          pk.x = pack2(o[d][4 * tq + 0] * inv, o[d][4 * tq + 1] * inv);
          pk.y = pack2(o[d][4 * tq + 2] * inv, o[d][4 * tq + 3] * inv);
          *reinterpret_cast<uint2*>(orow + dc) = pk;
        }
      }
  }
}

template <int HD>
int launch(const AttnArgs& a, int batch, int heads, hipStream_t st) {
  hipLaunchKernelGGL((k_attn<HD>), dim3((a.nq + QB - 1) / QB, heads, batch), dim3(THREADS), 0, st, a);
  return sdn_launch_status();
}

}  // namespace

extern "C" int sdn_attention_bf16(const void* q, const void* k, const void* v, void* out, int32_t batch,
                                  int32_t heads, int32_t nq, int32_t nk, int32_t head_dim, int32_t ldq,
                                  int32_t ldk, int32_t ldv, int32_t ldo, float scale, void* stream) {
  if (!q || !k || !v || !out || batch < 0 || heads <= 0 || nq <= 0 || nk <= 0) return SDN_E_INVALID;
  if ((nq & 31) || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3)) return SDN_E_INVALID;
  if (ldq < heads * head_dim || ldk < heads * head_dim || ldv < heads * head_dim || ldo < heads * head_dim)
    return SDN_E_INVALID;
  auto al = [](const void* p, int n) { return (reinterpret_cast<uintptr_t>(p) & (n - 1)) == 0; };
  if (!al(q, 16) || !al(k, 16) || !al(v, 16) || !al(out, 8)) return SDN_E_INVALID;
  if (batch == 0) return SDN_OK;
  AttnArgs a{(const unsigned short*)q, (const unsigned short*)k, (const unsigned short*)v, (unsigned short*)out,
             nq, nk, ldq, ldk, ldv, ldo, scale * 1.4426950408889634f};
  hipStream_t st = (hipStream_t)stream;
  switch (head_dim) {
    case 40: return launch<40>(a, batch, heads, st);
    case 64: return launch<64>(a, batch, heads, st);
    case 80: return launch<80>(a, batch, heads, st);
    case 160: return launch<160>(a, batch, heads, st);
    default: return SDN_E_INVALID;
  }
}
