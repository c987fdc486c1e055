// Groundwork for a precise GEMM at ~2 x (fp8 corrections) or ~1.5 x (fp6) the 16-bit cost instead of bf16x3's 3 x  (DESIGN 10.12).
//
// Scheme ("h8", emulated in tools/h8_emulation.py):  a w ~ fp16(a) fp16(w)  +  q8(2^11 (a - fp16(a))) q8(w) 2^-11  +  q8(a) q8(2^11 (w - fp16(w))) 2^-11
// with q8 = OCP e4m3.  On gfx950 the two correction products can run on v_mfma_scale_f32_16x16x128_f8f6f4 -- f32 accumulation into
// the SAME accumulators as the fp16 main term, the 2^-11 carried by the instruction's e8m0 block scale, K = 128 bytes per instruction
// (the byte geometry of a 64-element 16-bit k-tile, so the LDS-DMA tiles of k_gemm_dma carry either kind) at twice the bf16 rate.
//
// This program establishes, on the hardware, what such a kernel would build on:
//   1. LAYOUT + SCALE semantics of the instruction: C[16 x 16] = A[16 x 128] . B[16 x 128]^T from e4m3 bytes, against a host double
//      reference, for the hypothesised lane layout (lane l: row l & 15, bytes 32 (l >> 4) ... + 32), with scales 2^0 and 2^-11.
//   2. ACCURACY of the real instruction sequence on one 16 x 16 x K tile of f32 data with outlier channels: fp16 MFMA alone, the h8
//      sum, and bf16x3, against double.
//   3. RATE: the matrix-pipe time of one k-tile's worth of MFMAs of the 256 x 320 tile per wave (acc[4][10]), operands in registers,
//      2 waves per SIMD on all CUs:  16-bit (80 x 16x16x32) | bf16x3 (240) | h8 (80 + 40 x 16x16x128 fp8) | h6 (80 + 40 x fp6).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/h8_probe.hip -o /tmp/h8_probe && /tmp/h8_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- host-side number formats
static float e4m3_decode(unsigned char b) {                        // OCP e4m3fn: bias 7, no infinities, 0x7f / 0xff = NaN
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}
static unsigned char e4m3_encode(float x) {                        // round to nearest (ties away is fine for a probe), saturating
  if (x == 0.f || !isfinite(x)) return 0;
  const unsigned char s = x < 0 ? 0x80 : 0;
  float a = fabsf(x);
  if (a >= 448.f) return s | 0x7e;
  int e; float m = frexpf(a, &e);                                  // a = m 2^e, m in [0.5, 1)
  int E = e - 1 + 7;                                               // exponent field for 1.xxx 2^(e-1)
  if (E <= 0) { int q = (int)lrintf(ldexpf(a, 9)); if (q >= 8) return s | 0x08; return s | (unsigned char)q; }
  int q = (int)lrintf((m * 2.f - 1.f) * 8.f);
  if (q == 8) { q = 0; ++E; }
  if (E > 15 || (E == 15 && q == 7)) return s | 0x7e;
  return s | (unsigned char)(E << 3) | (unsigned char)q;
}
static unsigned short f32_to_bf16(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf16_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// ---------------------------------------------------------------- 1. layout probe
// lane l supplies 32 bytes of A and of B from lane-indexed buffers (the HOST decides which (row, k) those bytes are)
__global__ void k_probe(const int* a_lane, const int* b_lane, int scale_a, int scale_b, float* c_out) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = a_lane[l * 8 + i]; b[i] = b_lane[l * 8 + i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);     // cbsz = blgp = 0: e4m3 x e4m3
  for (int i = 0; i < 4; ++i) c_out[l * 4 + i] = c[i];
}

// ---------------------------------------------------------------- 2. one 16 x 16 x K tile: fp16 alone | h8 | bf16x3
// operands prepared on the host in the per-lane register order found by the probe
__global__ void k_tile(const f16x8* ah, const f16x8* wh, const i32x8* a8l, const i32x8* w8h, const i32x8* a8h, const i32x8* w8l,
                       const bf16x8* xa_hi, const bf16x8* xa_lo, const bf16x8* xw_hi, const bf16x8* xw_lo, int k16, int k8, float* out) {
  const int l = threadIdx.x;
  f32x4 c16 = {0.f, 0.f, 0.f, 0.f}, ch8, cx3 = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < k16; ++s) c16 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s * 64 + l], wh[s * 64 + l], c16, 0, 0, 0);
  ch8 = c16;
  const int s_m11 = 0x74747474, s_0 = 0x7f7f7f7f;                  // e8m0: 2^-11 = 127 - 11 = 116 = 0x74, 2^0 = 0x7f (all four scale bytes)
  for (int s = 0; s < k8; ++s) {
    ch8 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8l[s * 64 + l], w8h[s * 64 + l], ch8, 0, 0, 0, s_m11, 0, s_0);
    ch8 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8h[s * 64 + l], w8l[s * 64 + l], ch8, 0, 0, 0, s_0, 0, s_m11);
  }
  for (int s = 0; s < k16; ++s) {
    cx3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa_lo[s * 64 + l], xw_hi[s * 64 + l], cx3, 0, 0, 0);
    cx3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa_hi[s * 64 + l], xw_lo[s * 64 + l], cx3, 0, 0, 0);
    cx3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa_hi[s * 64 + l], xw_hi[s * 64 + l], cx3, 0, 0, 0);
  }
  for (int i = 0; i < 4; ++i) { out[l * 4 + i] = c16[i]; out[256 + l * 4 + i] = ch8[i]; out[512 + l * 4 + i] = cx3[i]; }
}

// ---------------------------------------------------------------- 3. rate: one k-tile's MFMAs of the 256 x 320 tile per wave
// MODE 0: 16-bit (2 k-steps x 4 x 10 x 16x16x32 f16) | 1: bf16x3 (3 x that, bf16) | 2: h8 (mode 0 + 4 x 10 scaled fp8) | 3: h6 (fp6 operands)
template <int MODE>
__global__ void __launch_bounds__(512) k_rate(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[4][10];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 10; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 fa[4], fw[10];
  bf16x8 ba[4], bw[10];
  i32x8 qa[1], qw[2];                                            // (a real kernel reads its fragments from LDS just in time; here two suffice)
  unsigned r = 0x9e3779b9u * (threadIdx.x + 1) + blockIdx.x;
  auto rnd = [&]() { r = r * 1664525u + 1013904223u; return r; };
  for (int i = 0; i < 4; ++i) { for (int e = 0; e < 8; ++e) { fa[i][e] = (_Float16)((int)(rnd() >> 24) / 64.f - 2.f); ba[i][e] = (__bf16)((int)(rnd() >> 24) / 64.f - 2.f); } }
  for (int e = 0; e < 8; ++e) { qa[0][e] = (int)(rnd() & 0x3f3f3f3f); qw[0][e] = (int)(rnd() & 0x3f3f3f3f); qw[1][e] = (int)(rnd() & 0x3f3f3f3f); }
  for (int j = 0; j < 10; ++j) { for (int e = 0; e < 8; ++e) { fw[j][e] = (_Float16)((int)(rnd() >> 24) / 64.f - 2.f); bw[j][e] = (__bf16)((int)(rnd() >> 24) / 64.f - 2.f); } }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 10; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[j], ba[i], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 10; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j], fa[i], acc[i][j], 0, 0, 0);
      if (MODE >= 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 10; ++j) {
            if (MODE == 2) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qw[j & 1], qa[0], acc[i][j], 0, 0, 0, 0x74747474, 0, 0x7f7f7f7f);
            else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(qw[j & 1], qa[0], acc[i][j], 2, 2, 0, 0x74747474, 0, 0x7f7f7f7f);   // e2m3 x e2m3
          }
      }
    }
    asm volatile("" : "+v"(acc[0][0]));                             // (the accumulator chains serialise the iterations; nothing to hoist)
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 10; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 512 + threadIdx.x] = s + (float)lane;
}

template <int MODE>
static double time_rate(float* d_out, int cus, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<MODE>, dim3(cus), dim3(512), 0, 0, d_out, 16);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_rate<MODE>, dim3(cus), dim3(512), 0, 0, d_out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e-3;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs\n", prop.gcnArchName, cus);

  // ---------------- 1. layout + scale
  srand(7);
  static unsigned char A8[16][128], B8[16][128];
  for (int m = 0; m < 16; ++m) for (int k = 0; k < 128; ++k) {
    do { A8[m][k] = (unsigned char)(rand() & 0xff); } while ((A8[m][k] & 0x7f) == 0x7f || (A8[m][k] & 0x78) > 0x50);      // |v| < 16: sums stay exact in f32
    do { B8[m][k] = (unsigned char)(rand() & 0xff); } while ((B8[m][k] & 0x7f) == 0x7f || (B8[m][k] & 0x78) > 0x50);
  }
  double ref[16][16];
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double s = 0; for (int k = 0; k < 128; ++k) s += (double)e4m3_decode(A8[m][k]) * e4m3_decode(B8[n][k]); ref[m][n] = s; }
  int *d_a, *d_b; float* d_c;
  CHECK(hipMalloc(&d_a, 64 * 32)); CHECK(hipMalloc(&d_b, 64 * 32)); CHECK(hipMalloc(&d_c, 64 * 16));
  int layout_ok = -1;
  for (int hyp = 0; hyp < 2 && layout_ok < 0; ++hyp) {
    static unsigned char la[64][32], lb[64][32];
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
      const int row = l & 15, q = l >> 4;
      const int k = hyp == 0 ? 32 * q + j : (j < 16 ? 16 * q + j : 64 + 16 * q + (j - 16));     // 0: 32 contiguous bytes; 1: two 16-byte halves, 64 apart
      la[l][j] = A8[row][k]; lb[l][j] = B8[row][k];
    }
    CHECK(hipMemcpy(d_a, la, sizeof la, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_b, lb, sizeof lb, hipMemcpyHostToDevice));
    for (int sc = 0; sc < 2; ++sc) {
      const int sa = sc == 0 ? 0x7f7f7f7f : 0x74747474;
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_a, d_b, sa, 0x7f7f7f7f, d_c);
      float c[64][4];
      CHECK(hipMemcpy(c, d_c, sizeof c, hipMemcpyDeviceToHost));
      double worst = 0, worst_t = 0;                                 // C[i] of lane l <-> (row 4 (l >> 4) + i, col l & 15), or its transpose
      for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        const double want = ref[4 * (l >> 4) + i][l & 15] * (sc == 0 ? 1.0 : ldexp(1.0, -11)), want_t = ref[l & 15][4 * (l >> 4) + i] * (sc == 0 ? 1.0 : ldexp(1.0, -11));
        worst = fmax(worst, fabs(c[l][i] - want)); worst_t = fmax(worst_t, fabs(c[l][i] - want_t));
      }
      printf("layout hypothesis %d (%s), scale_a = 2^%d: max |C - ref| = %.3e (C[i] = (4 (l >> 4) + i, l & 15)), %.3e (transposed)\n", hyp,
             hyp == 0 ? "lane l: row l & 15, bytes 32 (l >> 4) .. + 32" : "lane l: row l & 15, bytes 16 (l >> 4) .. + 16 and 64 + the same", sc == 0 ? 0 : -11, worst, worst_t);
      // (the k order inside a lane cannot be told apart -- a dot product does not care, as long as A and B agree -- what the probe pins is
      //  rows <-> lanes, the K split over the four lane groups, the C layout and the e8m0 scale; the unit's internal sum is not exact
      //  in f32: ~1e-5 of the result's magnitude)
      double mag = 0; for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) mag = fmax(mag, fabs(ref[m][n]));
      if (sc == 1 && worst < 1e-4 * mag * ldexp(1.0, -11)) layout_ok = hyp;
    }
  }
  printf("=> operand layout: hypothesis %d\n", layout_ok);

  // ---------------- 2. accuracy of the real instruction sequence on a 16 x 16 x K tile
  if (layout_ok >= 0) {
    const int K = 2880, k16 = K / 32, k8 = (K + 127) / 128, Kp = k8 * 128;
    static float Af[16][2944], Wf[16][2944];
    for (int m = 0; m < 16; ++m) for (int k = 0; k < Kp; ++k) {
      auto g = []() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); };
      Af[m][k] = k < K ? (float)(g() * ((k % 64) == 3 ? 12.0 : 1.0) * (0.5 + m / 8.0)) : 0.f;      // outlier channels, rows of different magnitude
      Wf[m][k] = k < K ? (float)(g() / sqrt((double)K)) : 0.f;
    }
    double refc[16][16];
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)Af[m][k] * Wf[n][k]; refc[m][n] = s; }
    // per-lane operand streams.  16x16x32 16-bit: lane l holds row l & 15, k = 32 s + 8 (l >> 4) + j
    _Float16* h_ah = (_Float16*)malloc(k16 * 64 * 16); _Float16* h_wh = (_Float16*)malloc(k16 * 64 * 16);
    unsigned short *h_xah = (unsigned short*)malloc(k16 * 64 * 16), *h_xal = (unsigned short*)malloc(k16 * 64 * 16), *h_xwh = (unsigned short*)malloc(k16 * 64 * 16), *h_xwl = (unsigned short*)malloc(k16 * 64 * 16);
    unsigned char *h_a8l = (unsigned char*)malloc(k8 * 64 * 32), *h_w8h = (unsigned char*)malloc(k8 * 64 * 32), *h_a8h = (unsigned char*)malloc(k8 * 64 * 32), *h_w8l = (unsigned char*)malloc(k8 * 64 * 32);
    for (int s = 0; s < k16; ++s) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
      const int row = l & 15, k = 32 * s + 8 * (l >> 4) + j, o = (s * 64 + l) * 8 + j;
      const float a = Af[row][k], w = Wf[row][k];
      h_ah[o] = (_Float16)a; h_wh[o] = (_Float16)w;
      const unsigned short ab = f32_to_bf16(a), wb = f32_to_bf16(w);
      h_xah[o] = ab; h_xwh[o] = wb; h_xal[o] = f32_to_bf16(a - bf16_to_f32(ab)); h_xwl[o] = f32_to_bf16(w - bf16_to_f32(wb));
    }
    for (int s = 0; s < k8; ++s) for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
      const int row = l & 15, q = l >> 4;
      const int k = 128 * s + (layout_ok == 0 ? 32 * q + j : (j < 16 ? 16 * q + j : 64 + 16 * q + (j - 16))), o = (s * 64 + l) * 32 + j;
      const float a = Af[row][k], w = Wf[row][k];
      const float ah = (float)(_Float16)a, wh = (float)(_Float16)w;
      h_a8l[o] = e4m3_encode(ldexpf(a - ah, 11)); h_a8h[o] = e4m3_encode(ah);
      h_w8l[o] = e4m3_encode(ldexpf(w - wh, 11)); h_w8h[o] = e4m3_encode(wh);
    }
    void *d[10]; const void* hs[10] = {h_ah, h_wh, h_a8l, h_w8h, h_a8h, h_w8l, h_xah, h_xal, h_xwh, h_xwl};
    const size_t sz[10] = {(size_t)k16 * 1024, (size_t)k16 * 1024, (size_t)k8 * 2048, (size_t)k8 * 2048, (size_t)k8 * 2048, (size_t)k8 * 2048, (size_t)k16 * 1024, (size_t)k16 * 1024, (size_t)k16 * 1024, (size_t)k16 * 1024};
    for (int i = 0; i < 10; ++i) { CHECK(hipMalloc(&d[i], sz[i])); CHECK(hipMemcpy(d[i], hs[i], sz[i], hipMemcpyHostToDevice)); }
    float* d_o; CHECK(hipMalloc(&d_o, 768 * 4));
    hipLaunchKernelGGL(k_tile, dim3(1), dim3(64), 0, 0, (const f16x8*)d[0], (const f16x8*)d[1], (const i32x8*)d[2], (const i32x8*)d[3], (const i32x8*)d[4], (const i32x8*)d[5],
                       (const bf16x8*)d[6], (const bf16x8*)d[7], (const bf16x8*)d[8], (const bf16x8*)d[9], k16, k8, d_o);
    float o[3][64][4];
    CHECK(hipMemcpy(o, d_o, sizeof o, hipMemcpyDeviceToHost));
    const char* nm[3] = {"fp16 MFMA alone", "h8: fp16 + two e4m3 corrections (scaled MFMA, 2^-11 in the block scale)", "bf16x3 (three bf16 MFMAs)"};
    for (int v = 0; v < 3; ++v) {
      double num[2] = {0, 0}, den = 0;
      for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        // 16-bit 16x16x32 with (A = first operand rows, B = second operand rows): C[i] of lane l <-> (4 (l >> 4) + i, l & 15)
        const double a0 = refc[4 * (l >> 4) + i][l & 15], a1 = refc[l & 15][4 * (l >> 4) + i];
        num[0] += (o[v][l][i] - a0) * (o[v][l][i] - a0); num[1] += (o[v][l][i] - a1) * (o[v][l][i] - a1); den += a0 * a0;
      }
      printf("accuracy, 16 x 16 x %d tile, outlier channels: %-74s rel L2 vs double %.3e\n", K, nm[v], sqrt(fmin(num[0], num[1]) / den));
    }
  }

  // ---------------- 3. rate
  float* d_out; CHECK(hipMalloc(&d_out, (size_t)cus * 512 * 4));
  const int iters = 4000;
  const double t16 = time_rate<0>(d_out, cus, iters), tx3 = time_rate<1>(d_out, cus, iters), th8 = time_rate<2>(d_out, cus, iters), th6 = time_rate<3>(d_out, cus, iters);
  const double flop16 = 2.0 * 80 * 16 * 16 * 32 * 8 * cus * (double)iters;           // algorithmic flop of the 16-bit k-tile (8 waves per CU)
  printf("rate (MFMAs of one k-tile of the 256 x 320 tile per wave, operands in registers, 8 waves per CU on %d CUs, %d iterations):\n", cus, iters);
  printf("  16-bit  (80 x 16x16x32 f16)                      %8.3f ms   %7.1f TFLOP/s algorithmic   x1.00\n", t16 * 1e3, flop16 / t16 / 1e12);
  printf("  bf16x3  (240 x 16x16x32 bf16)                    %8.3f ms   %7.1f TFLOP/s algorithmic   x%.2f\n", tx3 * 1e3, flop16 / tx3 / 1e12, tx3 / t16);
  printf("  h8      (80 f16 + 40 x 16x16x128 e4m3, scaled)   %8.3f ms   %7.1f TFLOP/s algorithmic   x%.2f\n", th8 * 1e3, flop16 / th8 / 1e12, th8 / t16);
  printf("  h6      (80 f16 + 40 x 16x16x128 e2m3, scaled)   %8.3f ms   %7.1f TFLOP/s algorithmic   x%.2f\n", th6 * 1e3, flop16 / th6 / 1e12, th6 / t16);
  return 0;
}
