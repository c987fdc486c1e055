// Shared device/host helpers for libsdn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sdn.h"

#define SDN_WAVE 64

static inline int sdn_launch_status() {
  return hipGetLastError() == hipSuccess ? SDN_OK : SDN_E_LAUNCH;
}

// Sum across the 64 lanes of a wave (xor butterfly: every lane ends with the total).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, SDN_WAVE);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, SDN_WAVE));
  return v;
}

// Sum across a workgroup of NW waves; `red` is an LDS array of >= NW floats. All threads get the total.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();                       // protect `red` from a previous use
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}

// ---- 16-bit storage types (bf16 = BASELINE's dtype; f16 = the reference's SD-v3 dtype / tight-parity mode) ----
typedef __attribute__((ext_vector_type(8))) __bf16 sdn_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 sdn_f16x8;
typedef __attribute__((ext_vector_type(4))) float sdn_f32x4;
typedef __attribute__((ext_vector_type(16))) float sdn_f32x16;

struct SdnBF16 {
  typedef sdn_bf16x8 v8;
  static constexpr int kDtype = 0;
  static __device__ __forceinline__ float to_f(unsigned v16) { return __uint_as_float(v16 << 16); }
  static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 v2;
    v2 p = {(__bf16)lo, (__bf16)hi};                      // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return *reinterpret_cast<unsigned*>(&p);
  }
  static __device__ __forceinline__ sdn_f32x4 mfma16(v8 a, v8 b, sdn_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  // acc + sum(v) / acc + sum(v * v) over the 8 elements (v_dot2c_f32_bf16)
  static __device__ __forceinline__ float dot_self(v8 v, float acc) {
    typedef __attribute__((ext_vector_type(2))) __bf16 v2;
    const v2* p = reinterpret_cast<const v2*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2_f32_bf16(p[i], p[i], acc, false);
    return acc;
  }
  static __device__ __forceinline__ float dot_ones(v8 v, float acc) {
    typedef __attribute__((ext_vector_type(2))) __bf16 v2;
    const v2* p = reinterpret_cast<const v2*>(&v);
    const v2 one = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2_f32_bf16(p[i], one, acc, false);
    return acc;
  }
  static __device__ __forceinline__ sdn_f32x16 mfma32(v8 a, v8 b, sdn_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
struct SdnF16 {
  typedef sdn_f16x8 v8;
  static constexpr int kDtype = 1;
  static __device__ __forceinline__ float to_f(unsigned v16) {
    unsigned short s = (unsigned short)v16;
    return (float)(*reinterpret_cast<_Float16*>(&s));
  }
  static __device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) _Float16 v2;
    v2 p = {(_Float16)lo, (_Float16)hi};                  // v_cvt_f16_f32 x2 (RNE) + pack
    return *reinterpret_cast<unsigned*>(&p);
  }
  static __device__ __forceinline__ sdn_f32x4 mfma16(v8 a, v8 b, sdn_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float dot_self(v8 v, float acc) {
    typedef __attribute__((ext_vector_type(2))) _Float16 v2;
    const v2* p = reinterpret_cast<const v2*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2(p[i], p[i], acc, false);
    return acc;
  }
  static __device__ __forceinline__ float dot_ones(v8 v, float acc) {
    typedef __attribute__((ext_vector_type(2))) _Float16 v2;
    const v2* p = reinterpret_cast<const v2*>(&v);
    const v2 one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2(p[i], one, acc, false);
    return acc;
  }
  static __device__ __forceinline__ sdn_f32x16 mfma32(v8 a, v8 b, sdn_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
