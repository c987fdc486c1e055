"""Which floating-point variant of Box-Muller reproduces THIS torch build's randn bit for bit?  (diagnostic for csrc/sdn_rng.hip)
Compiles a small kernel with hipcc on the GPU box and counts mismatches of 96 variants against torch.randn."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import torch

SRC = r'''
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#define INV 2.3283064e-10f
#define INV2PI 1.46291807e-09f
__global__ void k(unsigned long long seed, unsigned long long off, int mode, float* out, int n) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  rocrand_state_philox4x32_10 st;
  rocrand_init(seed, idx, off, &st);
  uint4 r = rocrand4(&st);
  unsigned x = r.x, y = r.y;
  float u = (mode & 1) ? fmaf((float)x, INV, INV) : __fadd_rn(INV, __fmul_rn((float)x, INV));
  float v = (mode & 2) ? fmaf((float)y, INV2PI, INV2PI) : __fadd_rn(INV2PI, __fmul_rn((float)y, INV2PI));
  float lg = (mode & 4) ? __logf(u) : logf(u);
  int sq = (mode >> 3) & 3;   // 0 sqrtf, 1 __fsqrt_rn, 2 native
  float arg = __fmul_rn(-2.0f, lg);
  float s = sq == 0 ? sqrtf(arg) : (sq == 1 ? __fsqrt_rn(arg) : __builtin_amdgcn_sqrtf(arg));
  float sn, cs;
  int sc = (mode >> 5) & 3;   // 0 __sincosf, 1 sincosf, 2 __sinf, 3 sinf
  if (sc == 0) __sincosf(v, &sn, &cs); else if (sc == 1) sincosf(v, &sn, &cs); else if (sc == 2) sn = __sinf(v); else sn = sinf(v);
  out[idx] = __fmul_rn(sn, s);
}
extern "C" void run(unsigned long long seed, unsigned long long off, int mode, float* out, int n) {
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, seed, off, mode, out, n);
  hipDeviceSynchronize();
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "v.hip"), "w").write(SRC)
subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(d, "v.hip"), "-o", os.path.join(d, "v.so")], check=True)
lib = C.CDLL(os.path.join(d, "v.so"))
lib.run.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_int]
n = 16384
g = torch.Generator(device="cuda").manual_seed(42)
ref = torch.randn(n, generator=g, device="cuda")
out = torch.empty(n, device="cuda")
best = []
for mode in range(128):
    if ((mode >> 3) & 3) == 3:
        continue
    lib.run(42, 0, mode, out.data_ptr(), n)
    best.append((int((out != ref).sum()), mode))
best.sort()
for nd, mode in best[:12]:
    print(f"mismatches {nd:6d}  mode {mode:3d}: u_fma={mode & 1} v_fma={(mode >> 1) & 1} fastlog={(mode >> 2) & 1} sqrt={(mode >> 3) & 3} sincos={(mode >> 5) & 3}")
