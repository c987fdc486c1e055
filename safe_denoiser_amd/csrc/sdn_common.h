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
