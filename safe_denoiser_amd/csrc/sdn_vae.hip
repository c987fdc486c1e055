// Small kernels around the VAE decoder plan (the convolutions, GroupNorms and linears of
// the decoder run on the GEMM / norm kernels of the UNet):
//   k_latent_mix    post_quant_conv: a 1x1 conv on the fp32 NCHW latent (C <= 16), with the 1/scaling_factor of
//                   StableDiffusionPipeline.decode_latents folded in;
//   k_softmax_rows  softmax(scale * S) over fp32 rows -> 16-bit P (the decoder's single-head, d = 512 mid-block
//                   attention is run as Q K^T GEMM -> row softmax -> P V GEMM: its head dim does not fit the flash
//                   kernel's register tile, and it is 1.5 % of the decoder's FLOPs);
//   k_transpose16   [R, C] -> [C, R] 16-bit (V^T as the "weight" operand of the P V GEMM);
//   k_image_post    (x / 2 + 0.5).clamp(0, 1) NCHW fp32 -> NHWC fp32 and / or round(255 x) uint8
//                   (decode_latents + numpy_to_pil of the reference's pipelines, ...threshold_time.py:589-596).
#include "sdn_common.h"

namespace {

constexpr int THREADS = 256;

__global__ void __launch_bounds__(THREADS)
k_latent_mix(const float* __restrict__ z, const float* __restrict__ w, const float* __restrict__ bias, int B, int C,
             int hw, float in_scale, float* __restrict__ out) {
  const long total = (long)B * hw;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += (long)gridDim.x * THREADS) {
    const long b = e / hw, p = e - b * hw;
    float v[16];
    for (int ci = 0; ci < C; ++ci) v[ci] = z[(b * C + ci) * hw + p] * in_scale;
    for (int co = 0; co < C; ++co) {
      float acc = bias[co];
      for (int ci = 0; ci < C; ++ci) acc = fmaf(w[co * C + ci], v[ci], acc);
      out[(b * C + co) * hw + p] = acc;
    }
  }
}

// One workgroup per row; the row (n <= 4 * NV * THREADS floats) is held in registers between the three sweeps.
template <typename T, int NV>
__global__ void __launch_bounds__(THREADS)
k_softmax_rows(const float* __restrict__ s, long ld_s, int n, float scale_log2e, unsigned short* __restrict__ out, long ld_o) {
  __shared__ float red[THREADS / 64];
  const float* row = s + (long)blockIdx.x * ld_s;
  unsigned short* orow = out + (long)blockIdx.x * ld_o;
  const int nv = n / 4;                                     // float4 groups
  float4 v[NV];
  float mx = -3.0e38f;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int i = threadIdx.x + q * THREADS;
    if (i < nv) {
      v[q] = *reinterpret_cast<const float4*>(row + 4 * i);
      mx = fmaxf(fmaxf(fmaxf(mx, v[q].x), fmaxf(v[q].y, v[q].z)), v[q].w);
    }
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float off = mx * scale_log2e;
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int i = threadIdx.x + q * THREADS;
    if (i < nv) {
      v[q].x = exp2f(fmaf(v[q].x, scale_log2e, -off)); v[q].y = exp2f(fmaf(v[q].y, scale_log2e, -off));
      v[q].z = exp2f(fmaf(v[q].z, scale_log2e, -off)); v[q].w = exp2f(fmaf(v[q].w, scale_log2e, -off));
      sum += (v[q].x + v[q].y) + (v[q].z + v[q].w);
    }
  }
  sum = block_sum<THREADS / 64>(sum, red);
  const float inv = 1.f / sum;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int i = threadIdx.x + q * THREADS;
    if (i < nv) {
      uint2 pk;
      pk.x = T::pack2(v[q].x * inv, v[q].y * inv);
      pk.y = T::pack2(v[q].z * inv, v[q].w * inv);
      *reinterpret_cast<uint2*>(orow + 4 * i) = pk;
    }
  }
}

// 64 x 64 tiles through LDS (+1 column pad: conflict-free both ways).
__global__ void __launch_bounds__(THREADS)
k_transpose16(const unsigned short* __restrict__ in, int R, int C, long ld_in, unsigned short* __restrict__ out, long ld_out) {
  __shared__ unsigned short tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += THREADS) {
    const int r = e >> 6, c = e & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? in[(long)(r0 + r) * ld_in + c0 + c] : (unsigned short)0;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += THREADS) {
    const int c = e >> 6, r = e & 63;
    if (r0 + r < R && c0 + c < C) out[(long)(c0 + c) * ld_out + r0 + r] = tile[r][c];
  }
}

__global__ void __launch_bounds__(THREADS)
k_image_post(const float* __restrict__ x, int B, int C, int hw, float* __restrict__ out01, unsigned char* __restrict__ out8) {
  const long total = (long)B * hw;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += (long)gridDim.x * THREADS) {
    const long b = e / hw, p = e - b * hw;
    for (int c = 0; c < C; ++c) {
      float v = x[(b * C + c) * hw + p] * 0.5f + 0.5f;
      v = fminf(fmaxf(v, 0.f), 1.f);
      if (!(v == v)) v = 0.f;                                // NaN -> 0 (torch.clamp keeps NaN; uint8 conversion of NaN is undefined)
      if (out01) out01[e * C + c] = v;
      if (out8) out8[e * C + c] = (unsigned char)rintf(v * 255.f);   // numpy_to_pil: (images * 255).round().astype(uint8)
    }
  }
}

// DiagonalGaussianDistribution.sample() * scale: moments [B, 2L, hw] = (mean | logvar), logvar clamped to [-30, 20].
__global__ void __launch_bounds__(THREADS)
k_gauss_sample(const float* __restrict__ mom, const float* __restrict__ noise, int B, int L, int hw, float scale,
               float* __restrict__ out) {
  const long total = (long)B * L * hw;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += (long)gridDim.x * THREADS) {
    const long b = e / ((long)L * hw), r = e - b * (long)L * hw;
    const float mean = mom[b * 2 * L * hw + r];
    float v = mean;
    if (noise) {
      const float lv = fminf(fmaxf(mom[b * 2 * L * hw + (long)L * hw + r], -30.f), 20.f);
      v = fmaf(expf(0.5f * lv), noise[e], mean);
    }
    out[e] = v * scale;
  }
}

// CLIPTextEmbeddings: out[b, t, :] = token_embedding[ids[b, t]] + position_embedding[t]   (16-bit tables, one rounding)
template <typename T>
__global__ void __launch_bounds__(THREADS)
k_clip_embed(const int* __restrict__ ids, const unsigned short* __restrict__ tok, const unsigned short* __restrict__ pos,
             long rows, int n, int C, int vocab, unsigned short* __restrict__ out) {
  const int cch = C / 2;
  const long total = rows * cch;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += (long)gridDim.x * THREADS) {
    const long r = e / cch; const int c = (int)(e - r * cch) * 2;
    int id = ids[r]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);       // out-of-range ids are clamped, not faulted on
    const unsigned a = *reinterpret_cast<const unsigned*>(tok + (long)id * C + c);
    const unsigned p = *reinterpret_cast<const unsigned*>(pos + (long)(r % n) * C + c);
    *reinterpret_cast<unsigned*>(out + r * C + c) = T::pack2(T::to_f(a & 0xffff) + T::to_f(p & 0xffff), T::to_f(a >> 16) + T::to_f(p >> 16));
  }
}

inline unsigned grid_for(long total) {
  long g = (total + THREADS - 1) / THREADS;
  return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int sdn_latent_mix(const float* z, const float* w, const float* bias, int32_t batch, int32_t channels,
                              int32_t hw, float in_scale, float* out, void* stream) {
  if (!z || !w || !bias || !out || batch < 0 || channels <= 0 || channels > 16 || hw <= 0) return SDN_E_INVALID;
  if (batch == 0) return SDN_OK;
  hipLaunchKernelGGL(k_latent_mix, dim3(grid_for((long)batch * hw)), dim3(THREADS), 0, (hipStream_t)stream, z, w, bias,
                     batch, channels, hw, in_scale, out);
  return sdn_launch_status();
}

extern "C" int sdn_softmax_rows(int32_t dtype, const float* scores, int64_t ld_scores, int64_t rows, int32_t n, float scale,
                                void* out, int64_t ld_out, void* stream) {
  if (!scores || !out || rows < 0 || n <= 0 || (n & 3) || n > 64 * THREADS || ld_scores < n || ld_out < n ||
      (ld_scores & 3) || (ld_out & 3) || (reinterpret_cast<uintptr_t>(scores) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & 7) || dtype < 0 || dtype > 1 || rows > 0x7fffffffL)
    return SDN_E_INVALID;
  if (rows == 0) return SDN_OK;
  const float sl = scale * 1.4426950408889634f;
#define SDN_SM_LAUNCH(TT, NV)                                                                                              \
  hipLaunchKernelGGL((k_softmax_rows<TT, NV>), dim3((unsigned)rows), dim3(THREADS), 0, (hipStream_t)stream, scores,           \
                     (long)ld_scores, n, sl, (unsigned short*)out, (long)ld_out)
  if (n <= 16 * THREADS) { if (dtype == 1) SDN_SM_LAUNCH(SdnF16, 4); else SDN_SM_LAUNCH(SdnBF16, 4); }
  else { if (dtype == 1) SDN_SM_LAUNCH(SdnF16, 16); else SDN_SM_LAUNCH(SdnBF16, 16); }
#undef SDN_SM_LAUNCH
  return sdn_launch_status();
}

extern "C" int sdn_transpose16(const void* in, int32_t rows, int32_t cols, int64_t ld_in, void* out, int64_t ld_out,
                               void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0 || ld_in < cols || ld_out < rows) return SDN_E_INVALID;
  hipLaunchKernelGGL(k_transpose16, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(THREADS), 0, (hipStream_t)stream,
                     (const unsigned short*)in, rows, cols, (long)ld_in, (unsigned short*)out, (long)ld_out);
  return sdn_launch_status();
}

extern "C" int sdn_image_postprocess(const float* image_nchw, int32_t batch, int32_t channels, int32_t height,
                                     int32_t width, float* out_nhwc01, uint8_t* out_nhwc_u8, void* stream) {
  if (!image_nchw || (!out_nhwc01 && !out_nhwc_u8) || batch < 0 || channels <= 0 || height <= 0 || width <= 0)
    return SDN_E_INVALID;
  if (batch == 0) return SDN_OK;
  const int hw = height * width;
  hipLaunchKernelGGL(k_image_post, dim3(grid_for((long)batch * hw)), dim3(THREADS), 0, (hipStream_t)stream, image_nchw,
                     batch, channels, hw, out_nhwc01, out_nhwc_u8);
  return sdn_launch_status();
}

extern "C" int sdn_gaussian_sample(const float* moments, const float* noise, int32_t batch, int32_t latent_channels,
                                   int32_t hw, float scale, float* out, void* stream) {
  if (!moments || !out || batch < 0 || latent_channels <= 0 || hw <= 0) return SDN_E_INVALID;
  if (batch == 0) return SDN_OK;
  hipLaunchKernelGGL(k_gauss_sample, dim3(grid_for((long)batch * latent_channels * hw)), dim3(THREADS), 0, (hipStream_t)stream,
                     moments, noise, batch, latent_channels, hw, scale, out);
  return sdn_launch_status();
}

extern "C" int sdn_clip_embed(int32_t dtype, const int32_t* input_ids, const void* token_embedding, const void* position_embedding,
                              int64_t rows, int32_t seq_len, int32_t hidden, int32_t vocab, void* out, void* stream) {
  if (!input_ids || !token_embedding || !position_embedding || !out || rows < 0 || seq_len <= 0 || hidden <= 0 || (hidden & 1) ||
      vocab <= 0 || dtype < 0 || dtype > 1)
    return SDN_E_INVALID;
  if (rows == 0) return SDN_OK;
  if (dtype == 1)
    hipLaunchKernelGGL((k_clip_embed<SdnF16>), dim3(grid_for(rows * (hidden / 2))), dim3(THREADS), 0, (hipStream_t)stream, input_ids,
                       (const unsigned short*)token_embedding, (const unsigned short*)position_embedding, (long)rows, seq_len, hidden,
                       vocab, (unsigned short*)out);
  else
    hipLaunchKernelGGL((k_clip_embed<SdnBF16>), dim3(grid_for(rows * (hidden / 2))), dim3(THREADS), 0, (hipStream_t)stream, input_ids,
                       (const unsigned short*)token_embedding, (const unsigned short*)position_embedding, (long)rows, seq_len, hidden,
                       vocab, (unsigned short*)out);
  return sdn_launch_status();
}
