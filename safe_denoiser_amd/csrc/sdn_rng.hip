// Batched standard-normal draws that reproduce `torch.randn(shape, generator=g, device="cuda", dtype=float32)` bit for bit,
// for P independent generators in ONE launch (every prompt keeps its own torch.Generator so that its random
// stream -- latents, the x0 probe's discarded variance draw, the conditional re-noise draw, the step's variance draw -- is the
// reference's; run_nudity.py:142,448).  The per-prompt loop of `torch.randn` + slice copy was 3P launches + 3P copies per
// step; this is one launch per draw kind.
//
// What torch does for a float tensor of `numel` elements (ATen/native/cuda/DistributionTemplates.h, normal_ -> normal_and_
// transform -> distribution_nullary_kernel with unroll_factor 4): block 256, grid = min(multiProcessorCount *
// (maxThreadsPerMultiProcessor / 256), ceil(numel / 256)); thread idx initialises Philox4x32-10 with (seed, subsequence = idx,
// offset = the generator's philox offset), and per pass over `linear = idx; linear < rounded; linear += 256 * grid * 4` draws ONE
// normal4 (Box-Muller on the four 32-bit outputs) and stores component ii at linear + 256 * grid * ii when that is < numel.
// The generator then advances by ((numel - 1) / (256 * grid * 4) + 1) * 4.  Philox comes from rocRAND's device API (what
// hipRAND wraps); the Box-Muller step is restated with its floating-point contraction made EXPLICIT: torch's build contracts the
// two uniform mappings u = 2^-32 + x 2^-32 and v = 2^-32 2pi + y 2^-32 2pi into FMAs and nothing else (found by running the 96
// variants of tools/rng_variants.py against torch.randn on the MI355X box: only the variants with the FMA in v reproduce
// torch bit for bit; the library itself is built with -ffp-contract=off).  tests/test_gpu_rng.py holds the bit-equality.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>

#include "sdn_common.h"

namespace {

// (sin, cos)(2 pi v) * sqrt(-2 ln u): rocrand_normal.h's box_muller with torch's contraction (see the header comment)
__device__ __forceinline__ float2 box_muller_as_torch(unsigned x, unsigned y) {
  const float u = fmaf((float)x, ROCRAND_2POW32_INV, ROCRAND_2POW32_INV);
  const float v = fmaf((float)y, ROCRAND_2POW32_INV_2PI, ROCRAND_2POW32_INV_2PI);
  const float s = sqrtf(-2.0f * logf(u));
  float2 r;
  __sincosf(v, &r.x, &r.y);
  r.x *= s;
  r.y *= s;
  return r;
}

__global__ void __launch_bounds__(256)
k_randn_philox(const unsigned long long* __restrict__ seeds, const unsigned long long* __restrict__ offsets,
               const int* __restrict__ rows, long numel, int grid_t, float* __restrict__ out) {
  // blockIdx.y = generator; blockIdx.x = torch's block index for that generator's own launch
  const int p = blockIdx.y;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = 256L * grid_t;                                        // torch's blockDim.x * gridDim.x
  rocrand_state_philox4x32_10 st;
  rocrand_init(seeds[p], (unsigned long long)idx, offsets[p], &st);
  float* dst = out + (long)(rows ? rows[p] : p) * numel;
  const long rounded = ((numel - 1) / (stride * 4) + 1) * stride * 4;
  for (long linear = idx; linear < rounded; linear += stride * 4) {
    const uint4 r = rocrand4(&st);
    const float2 a = box_muller_as_torch(r.x, r.y), b = box_muller_as_torch(r.z, r.w);
    const float v[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const long li = linear + stride * ii;
      if (li < numel) dst[li] = v[ii];
    }
  }
}

// device-resident generator states: the same draw, for the rows whose flag is set (null = all), into row p itself
__global__ void __launch_bounds__(256)
k_randn_philox_state(const unsigned long long* __restrict__ seeds, const unsigned long long* __restrict__ offsets,
                     const int* __restrict__ flags, long numel, int grid_t, float* __restrict__ out) {
  const int p = blockIdx.y;
  if (flags && flags[p] == 0) return;                                       // block-uniform: this prompt draws nothing
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = 256L * grid_t;
  rocrand_state_philox4x32_10 st;
  rocrand_init(seeds[p], (unsigned long long)idx, offsets[p], &st);
  float* dst = out + (long)p * numel;
  const long rounded = ((numel - 1) / (stride * 4) + 1) * stride * 4;
  for (long linear = idx; linear < rounded; linear += stride * 4) {
    const uint4 r = rocrand4(&st);
    const float2 a = box_muller_as_torch(r.x, r.y), b = box_muller_as_torch(r.z, r.w);
    const float v[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const long li = linear + stride * ii;
      if (li < numel) dst[li] = v[ii];
    }
  }
}

// a separate launch: every block of the draw has read offsets[p] by the time the stream reaches this kernel
__global__ void k_philox_advance(unsigned long long* __restrict__ offsets, const int* __restrict__ flags, int n_gen,
                                 unsigned long long inc) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_gen && (!flags || flags[p] != 0)) offsets[p] += inc;
}

}  // namespace

// torch's launch geometry for `numel` elements on the current device: grid size, and the philox offset increment.
extern "C" int sdn_randn_philox_plan(int64_t numel, int32_t* grid_out, int64_t* offset_increment_out) {
  if (numel <= 0) return SDN_E_INVALID;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return SDN_E_LAUNCH;
  const unsigned long long blocks = ((unsigned long long)numel + 255) / 256;
  const unsigned long long cap = (unsigned long long)prop.multiProcessorCount * (prop.maxThreadsPerMultiProcessor / 256);
  const unsigned long long grid = blocks < cap ? blocks : cap;
  if (grid_out) *grid_out = (int32_t)grid;
  if (offset_increment_out) *offset_increment_out = (int64_t)((((unsigned long long)numel - 1) / (256ULL * grid * 4) + 1) * 4);
  return SDN_OK;
}

// out[rows[p] (or p), 0 .. numel) = torch.randn(numel, generator(seed = seeds[p], philox offset = offsets[p])) for p < n_gen.
// seeds / offsets / rows are DEVICE arrays (rows may be null = identity).  The caller advances every generator by the
// increment sdn_randn_philox_plan reports.
extern "C" int sdn_randn_philox(const uint64_t* seeds, const uint64_t* offsets, const int32_t* rows, int32_t n_gen, int64_t numel,
                                float* out, void* stream) {
  if (!seeds || !offsets || !out || n_gen < 0 || numel <= 0 || n_gen > 65535) return SDN_E_INVALID;
  if (n_gen == 0) return SDN_OK;
  int32_t grid = 0;
  const int rc = sdn_randn_philox_plan(numel, &grid, nullptr);
  if (rc != SDN_OK) return rc;
  hipLaunchKernelGGL(k_randn_philox, dim3((unsigned)grid, (unsigned)n_gen), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned long long*)seeds, (const unsigned long long*)offsets, rows, (long)numel, (int)grid, out);
  return sdn_launch_status();
}

// The same draw with the generators' (seed, philox offset) pairs RESIDENT on the device: row p of `out` receives its normals when
// flags == null or flags[p] != 0 (flags: the loop's device-side is_negation vector -- the conditional re-noise draw needs no
// host-built index list), then offsets[p] advances by the plan's increment ON THE DEVICE.  out == null: advance only (the x0
// probe's discarded scheduler.step draw).  No host -> device traffic per draw; the caller mirrors the advance on its
// torch.Generator objects (set_offset) from the flags it has read back anyway.
extern "C" int sdn_randn_philox_state(const uint64_t* seeds, uint64_t* offsets, const int32_t* flags, int32_t n_gen, int64_t numel,
                                      float* out, void* stream) {
  if (!seeds || !offsets || n_gen < 0 || numel <= 0 || n_gen > 65535) return SDN_E_INVALID;
  if (n_gen == 0) return SDN_OK;
  int32_t grid = 0;
  int64_t inc = 0;
  const int rc = sdn_randn_philox_plan(numel, &grid, &inc);
  if (rc != SDN_OK) return rc;
  if (out)
    hipLaunchKernelGGL(k_randn_philox_state, dim3((unsigned)grid, (unsigned)n_gen), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)seeds, (const unsigned long long*)offsets, flags, (long)numel, (int)grid, out);
  hipLaunchKernelGGL(k_philox_advance, dim3((unsigned)((n_gen + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long*)offsets, flags, (int)n_gen, (unsigned long long)inc);
  return sdn_launch_status();
}
