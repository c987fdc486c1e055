// Guidance combine + scheduler step math: fp32 elementwise, HBM-bound, float4 grid-stride kernels.
// Rows P2/P3/S1/S1'/S3 of SURVEY.md section 8a.  Every kernel is one pass: read operands once, write once.
#include "sdn_common.h"

namespace {

constexpr int kThreads = 256;
inline int grid_for(int64_t n_vec) {
  int64_t g = (n_vec + kThreads - 1) / kThreads;
  if (g > 2048) g = 2048;            // cap + grid-stride (guide: Guideline 11)
  if (g < 1) g = 1;
  return (int)g;
}

__global__ void __launch_bounds__(kThreads)
k_cfg_combine(const float4* __restrict__ mo, int64_t pd4, float g, float4* __restrict__ eps) {
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < pd4; i += (int64_t)gridDim.x * kThreads) {
    float4 u = mo[i], t = mo[pd4 + i], o;
    o.x = u.x + g * (t.x - u.x); o.y = u.y + g * (t.y - u.y);
    o.z = u.z + g * (t.z - u.z); o.w = u.w + g * (t.w - u.w);
    eps[i] = o;
  }
}

// Safe-latent-diffusion guidance (eq. 3-8), one pass: reads the three branches + the momentum state, writes eps and the
// new momentum.  Elementwise, so the float4 is processed lane-wise.
__device__ __forceinline__ float sld_one(float u, float t, float c, float& mom, float g, float sg, float thr, float ms,
                                         float mb, int apply) {
  float guide = t - u;
  float scale = fminf(fabsf(t - c) * sg, 1.f);                  // eq. 6
  scale = (t - c) >= thr ? 0.f : scale;
  float gs = (c - u) * scale;                                   // eq. 4
  gs = gs + ms * mom;                                           // eq. 7
  mom = mb * mom + (1.f - mb) * gs;                             // eq. 8
  if (apply) guide -= gs;                                       // eq. 3 (after the warm-up)
  return u + g * guide;
}

__global__ void __launch_bounds__(kThreads)
k_sld_guidance(const float4* __restrict__ mo, int64_t pd4, float g, float sg, float thr, float ms, float mb, int apply,
               float4* __restrict__ mom, float4* __restrict__ eps) {
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < pd4; i += (int64_t)gridDim.x * kThreads) {
    const float4 u = mo[i], t = mo[pd4 + i], c = mo[2 * pd4 + i];
    float4 m = mom[i], o;
    o.x = sld_one(u.x, t.x, c.x, m.x, g, sg, thr, ms, mb, apply);
    o.y = sld_one(u.y, t.y, c.y, m.y, g, sg, thr, ms, mb, apply);
    o.z = sld_one(u.z, t.z, c.z, m.z, g, sg, thr, ms, mb, apply);
    o.w = sld_one(u.w, t.w, c.w, m.w, g, sg, thr, ms, mb, apply);
    mom[i] = m; eps[i] = o;
  }
}

// per-prompt guidance scale (a prompt table with a `guidance` column, run_nudity.py:390-396): blockIdx.y = prompt
__global__ void __launch_bounds__(kThreads)
k_cfg_combine_rows(const float4* __restrict__ mo, int64_t pd4, int64_t d4, const float* __restrict__ g_rows,
                   float4* __restrict__ eps) {
  const int p = blockIdx.y;
  const float g = g_rows[p];
  const int64_t base = (int64_t)p * d4;
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < d4; i += (int64_t)gridDim.x * kThreads) {
    float4 u = mo[base + i], t = mo[pd4 + base + i], o;
    o.x = u.x + g * (t.x - u.x); o.y = u.y + g * (t.y - u.y);
    o.z = u.z + g * (t.z - u.z); o.w = u.w + g * (t.w - u.w);
    eps[base + i] = o;
  }
}

__global__ void __launch_bounds__(kThreads)
k_sld_guidance_rows(const float4* __restrict__ mo, int64_t pd4, int64_t d4, const float* __restrict__ g_rows, float sg,
                    float thr, float ms, float mb, int apply, float4* __restrict__ mom, float4* __restrict__ eps) {
  const int p = blockIdx.y;
  const float g = g_rows[p];
  const int64_t base = (int64_t)p * d4;
  for (int64_t j = blockIdx.x * (int64_t)kThreads + threadIdx.x; j < d4; j += (int64_t)gridDim.x * kThreads) {
    const int64_t i = base + j;
    const float4 u = mo[i], t = mo[pd4 + i], c = mo[2 * pd4 + i];
    float4 m = mom[i], o;
    o.x = sld_one(u.x, t.x, c.x, m.x, g, sg, thr, ms, mb, apply);
    o.y = sld_one(u.y, t.y, c.y, m.y, g, sg, thr, ms, mb, apply);
    o.z = sld_one(u.z, t.z, c.z, m.z, g, sg, thr, ms, mb, apply);
    o.w = sld_one(u.w, t.w, c.w, m.w, g, sg, thr, ms, mb, apply);
    mom[i] = m; eps[i] = o;
  }
}

__device__ __forceinline__ float x0_of(float x, float e, float sa, float s1) { return (x - s1 * e) / sa; }
__device__ __forceinline__ float clampf(float v, float c) { return c > 0.f ? fminf(fmaxf(v, -c), c) : v; }

__global__ void __launch_bounds__(kThreads)
k_pred_x0(const float4* __restrict__ x, const float4* __restrict__ e, int64_t n4, float sa, float s1, float clip,
          float4* __restrict__ x0) {
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
    float4 a = x[i], b = e[i], o;
    o.x = clampf(x0_of(a.x, b.x, sa, s1), clip); o.y = clampf(x0_of(a.y, b.y, sa, s1), clip);
    o.z = clampf(x0_of(a.z, b.z, sa, s1), clip); o.w = clampf(x0_of(a.w, b.w, sa, s1), clip);
    x0[i] = o;
  }
}

template <bool HAS_NOISE>
__global__ void __launch_bounds__(kThreads)
k_sched_step(const float4* __restrict__ x, const float4* __restrict__ e, const float4* __restrict__ z, int64_t n4,
             float sa, float s1, float c0, float cx, float ce, float sg, float clip, float4* __restrict__ prev) {
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
    float4 a = x[i], b = e[i], o;
    float4 nz = HAS_NOISE ? z[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    o.x = c0 * clampf(x0_of(a.x, b.x, sa, s1), clip) + cx * a.x + ce * b.x + sg * nz.x;
    o.y = c0 * clampf(x0_of(a.y, b.y, sa, s1), clip) + cx * a.y + ce * b.y + sg * nz.y;
    o.z = c0 * clampf(x0_of(a.z, b.z, sa, s1), clip) + cx * a.z + ce * b.z + sg * nz.z;
    o.w = c0 * clampf(x0_of(a.w, b.w, sa, s1), clip) + cx * a.w + ce * b.w + sg * nz.w;
    prev[i] = o;
  }
}

__global__ void __launch_bounds__(kThreads)
k_axpby(const float4* __restrict__ a, const float4* __restrict__ b, int64_t n4, float ca, float cb,
        float4* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
    float4 p = a[i], q = b[i], o;
    o.x = ca * p.x + cb * q.x; o.y = ca * p.y + cb * q.y; o.z = ca * p.z + cb * q.z; o.w = ca * p.w + cb * q.w;
    out[i] = o;
  }
}

__global__ void __launch_bounds__(kThreads)
k_renoise_select(float4* __restrict__ lat, const float4* __restrict__ x0r, const float4* __restrict__ z,
                 const int32_t* __restrict__ isneg, int64_t d4, float sa, float s1) {
  const int p = blockIdx.y;
  if (isneg[p] == 0) return;                      // wave-uniform: whole row keeps its latents
  const int64_t base = (int64_t)p * d4;
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < d4; i += (int64_t)gridDim.x * kThreads) {
    float4 a = x0r[base + i], n = z[base + i], o;
    o.x = sa * a.x + s1 * n.x; o.y = sa * a.y + s1 * n.y; o.z = sa * a.z + s1 * n.z; o.w = sa * a.w + s1 * n.w;
    lat[base + i] = o;
  }
}

__global__ void __launch_bounds__(kThreads)
k_flow_endpoints(const float4* __restrict__ x, const float4* __restrict__ v, int64_t n4, float sigma,
                 float4* __restrict__ x0, float4* __restrict__ x1) {
  const float om = 1.f - sigma;
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
    float4 a = x[i], b = v[i], p, q;
    p.x = a.x - sigma * b.x; p.y = a.y - sigma * b.y; p.z = a.z - sigma * b.z; p.w = a.w - sigma * b.w;
    q.x = a.x + om * b.x; q.y = a.y + om * b.y; q.z = a.z + om * b.z; q.w = a.w + om * b.w;
    x0[i] = p; x1[i] = q;
  }
}

__global__ void __launch_bounds__(kThreads)
k_flow_renoise(const float4* __restrict__ x0r, const float4* __restrict__ x1, const float4* __restrict__ z,
               int64_t n4, float sn, float sq_sn, float sq_1m, float4* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
    float4 a = x0r[i], b = x1[i], c = z[i], o;
    float t;
    t = sq_sn * b.x + sq_1m * c.x; o.x = a.x + sn * (t - a.x);
    t = sq_sn * b.y + sq_1m * c.y; o.y = a.y + sn * (t - a.y);
    t = sq_sn * b.z + sq_1m * c.z; o.z = a.z + sn * (t - a.z);
    t = sq_sn * b.w + sq_1m * c.w; o.w = a.w + sn * (t - a.w);
    out[i] = o;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int sdn_cfg_combine(const float* model_out, int32_t n_prompt, int32_t n_branch, int64_t d, float g, float* eps,
                    void* stream) {
  if (!model_out || !eps || n_prompt < 0 || (n_branch != 2 && n_branch != 3) || d < 0 || (d & 3) ||
      !aligned16(model_out) || !aligned16(eps))
    return SDN_E_INVALID;
  const int64_t pd4 = (int64_t)n_prompt * d / 4;
  if (pd4 == 0) return SDN_OK;
  hipLaunchKernelGGL(k_cfg_combine, dim3(grid_for(pd4)), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float4*)model_out, pd4, g, (float4*)eps);
  return sdn_launch_status();
}

int sdn_sld_guidance(const float* model_out, int32_t n_prompt, int64_t d, float guidance_scale, float sld_guidance_scale,
                     float sld_threshold, float sld_momentum_scale, float sld_mom_beta, int32_t apply_safety,
                     float* momentum, float* eps, void* stream) {
  if (!model_out || !momentum || !eps || n_prompt < 0 || d < 0 || (d & 3) || !aligned16(model_out) ||
      !aligned16(momentum) || !aligned16(eps))
    return SDN_E_INVALID;
  const int64_t pd4 = (int64_t)n_prompt * d / 4;
  if (pd4 == 0) return SDN_OK;
  hipLaunchKernelGGL(k_sld_guidance, dim3(grid_for(pd4)), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float4*)model_out, pd4, guidance_scale, sld_guidance_scale, sld_threshold, sld_momentum_scale,
                     sld_mom_beta, apply_safety, (float4*)momentum, (float4*)eps);
  return sdn_launch_status();
}

int sdn_cfg_combine_rows(const float* model_out, int32_t n_prompt, int32_t n_branch, int64_t d, const float* guidance_rows,
                         float* eps, void* stream) {
  if (!model_out || !eps || !guidance_rows || n_prompt < 0 || n_prompt > 65535 || (n_branch != 2 && n_branch != 3) || d < 0 ||
      (d & 3) || !aligned16(model_out) || !aligned16(eps))
    return SDN_E_INVALID;
  const int64_t d4 = d / 4;
  if (n_prompt == 0 || d4 == 0) return SDN_OK;
  int gx = grid_for(d4);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_cfg_combine_rows, dim3(gx, n_prompt), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float4*)model_out, (int64_t)n_prompt * d4, d4, guidance_rows, (float4*)eps);
  return sdn_launch_status();
}

int sdn_sld_guidance_rows(const float* model_out, int32_t n_prompt, int64_t d, const float* guidance_rows,
                          float sld_guidance_scale, float sld_threshold, float sld_momentum_scale, float sld_mom_beta,
                          int32_t apply_safety, float* momentum, float* eps, void* stream) {
  if (!model_out || !momentum || !eps || !guidance_rows || n_prompt < 0 || n_prompt > 65535 || d < 0 || (d & 3) ||
      !aligned16(model_out) || !aligned16(momentum) || !aligned16(eps))
    return SDN_E_INVALID;
  const int64_t d4 = d / 4;
  if (n_prompt == 0 || d4 == 0) return SDN_OK;
  int gx = grid_for(d4);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_sld_guidance_rows, dim3(gx, n_prompt), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float4*)model_out, (int64_t)n_prompt * d4, d4, guidance_rows, sld_guidance_scale, sld_threshold,
                     sld_momentum_scale, sld_mom_beta, apply_safety, (float4*)momentum, (float4*)eps);
  return sdn_launch_status();
}

int sdn_pred_x0(const float* x, const float* eps, int64_t n, float sa, float s1, float clip, float* x0,
                void* stream) {
  if (!x || !eps || !x0 || n < 0 || (n & 3) || !aligned16(x) || !aligned16(eps) || !aligned16(x0) || sa == 0.f)
    return SDN_E_INVALID;
  if (n == 0) return SDN_OK;
  hipLaunchKernelGGL(k_pred_x0, dim3(grid_for(n / 4)), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float4*)x, (const float4*)eps, n / 4, sa, s1, clip, (float4*)x0);
  return sdn_launch_status();
}

int sdn_sched_step(const float* x, const float* eps, const float* noise, int64_t n, float sa, float s1, float c0,
                   float cx, float ce, float sg, float clip, float* prev, void* stream) {
  if (!x || !eps || !prev || n < 0 || (n & 3) || !aligned16(x) || !aligned16(eps) || !aligned16(prev) ||
      (noise && !aligned16(noise)) || sa == 0.f || (!noise && sg != 0.f))
    return SDN_E_INVALID;
  if (n == 0) return SDN_OK;
  if (noise)
    hipLaunchKernelGGL(k_sched_step<true>, dim3(grid_for(n / 4)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const float4*)x, (const float4*)eps, (const float4*)noise, n / 4, sa, s1, c0, cx, ce, sg, clip,
                       (float4*)prev);
  else
    hipLaunchKernelGGL(k_sched_step<false>, dim3(grid_for(n / 4)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const float4*)x, (const float4*)eps, (const float4*)nullptr, n / 4, sa, s1, c0, cx, ce, 0.f,
                       clip, (float4*)prev);
  return sdn_launch_status();
}

int sdn_add_noise(const float* x0, const float* noise, int64_t n, float sa, float s1, float* out, void* stream) {
  if (!x0 || !noise || !out || n < 0 || (n & 3) || !aligned16(x0) || !aligned16(noise) || !aligned16(out))
    return SDN_E_INVALID;
  if (n == 0) return SDN_OK;
  hipLaunchKernelGGL(k_axpby, dim3(grid_for(n / 4)), dim3(kThreads), 0, (hipStream_t)stream, (const float4*)x0,
                     (const float4*)noise, n / 4, sa, s1, (float4*)out);
  return sdn_launch_status();
}

int sdn_renoise_select(float* latents, const float* x0r, const float* noise, const int32_t* isneg, int32_t n_prompt,
                       int64_t d, float sa, float s1, void* stream) {
  if (!latents || !x0r || !noise || !isneg || n_prompt < 0 || d < 0 || (d & 3) || !aligned16(latents) ||
      !aligned16(x0r) || !aligned16(noise))
    return SDN_E_INVALID;
  if (n_prompt == 0 || d == 0) return SDN_OK;
  int gx = grid_for(d / 4);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_renoise_select, dim3(gx, n_prompt), dim3(kThreads), 0, (hipStream_t)stream, (float4*)latents,
                     (const float4*)x0r, (const float4*)noise, isneg, d / 4, sa, s1);
  return sdn_launch_status();
}

int sdn_flow_euler_step(const float* x, const float* v, int64_t n, float sigma, float sigma_next, float* prev,
                        void* stream) {
  if (!x || !v || !prev || n < 0 || (n & 3) || !aligned16(x) || !aligned16(v) || !aligned16(prev))
    return SDN_E_INVALID;
  if (n == 0) return SDN_OK;
  hipLaunchKernelGGL(k_axpby, dim3(grid_for(n / 4)), dim3(kThreads), 0, (hipStream_t)stream, (const float4*)x,
                     (const float4*)v, n / 4, 1.f, sigma_next - sigma, (float4*)prev);
  return sdn_launch_status();
}

int sdn_flow_endpoints(const float* x, const float* v, int64_t n, float sigma, float* x0, float* x1, void* stream) {
  if (!x || !v || !x0 || !x1 || n < 0 || (n & 3) || !aligned16(x) || !aligned16(v) || !aligned16(x0) ||
      !aligned16(x1))
    return SDN_E_INVALID;
  if (n == 0) return SDN_OK;
  hipLaunchKernelGGL(k_flow_endpoints, dim3(grid_for(n / 4)), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float4*)x, (const float4*)v, n / 4, sigma, (float4*)x0, (float4*)x1);
  return sdn_launch_status();
}

int sdn_flow_renoise(const float* x0r, const float* x1, const float* z, int64_t n, float sigma_next, float* out,
                     void* stream) {
  if (!x0r || !x1 || !z || !out || n < 0 || (n & 3) || sigma_next < 0.f || sigma_next > 1.f || !aligned16(x0r) ||
      !aligned16(x1) || !aligned16(z) || !aligned16(out))
    return SDN_E_INVALID;
  if (n == 0) return SDN_OK;
  hipLaunchKernelGGL(k_flow_renoise, dim3(grid_for(n / 4)), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float4*)x0r, (const float4*)x1, (const float4*)z, n / 4, sigma_next, sqrtf(sigma_next),
                     sqrtf(1.f - sigma_next), (float4*)out);
  return sdn_launch_status();
}

}  // extern "C"

namespace {
__global__ void __launch_bounds__(256) k_repeat(const uint4* __restrict__ in, long n16, int rep, uint4* __restrict__ out) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n16; e += (long)gridDim.x * 256) {
    const uint4 v = in[e];
    for (int k = 0; k < rep; ++k) out[(long)k * n16 + e] = v;
  }
}
}  // namespace

extern "C" int sdn_repeat(const void* in, size_t bytes, int32_t rep, void* out, void* stream) {
  if (!in || !out || rep <= 0 || (bytes & 15) || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
    return SDN_E_INVALID;
  if (bytes == 0) return SDN_OK;
  const long n16 = (long)(bytes / 16);
  long grid = (n16 + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_repeat, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, n16, rep, (uint4*)out);
  return sdn_launch_status();
}
