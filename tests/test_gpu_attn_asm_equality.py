"""The attention kernels' inline-asm `ds_read_b64_tr_b16` pipeline vs the compiler's builtin reads: the test builds
csrc/sdn_attn.hip a second time with -DSDN_ATTN_NO_ASM_TR (hipcc is on the GPU box; ~10 s) and runs both builds on the same inputs
for every head dim, both storage types, the two-query-set variant (d = 40, long key sets) and a ragged short key set: the outputs
must be BIT-identical -- the asm form only changes when LDS reads are issued and waited for, never what is computed."""
import ctypes as C
import os
import shutil
import subprocess

import pytest
import torch

import safe_denoiser_amd as sda
from safe_denoiser_amd import _lib

pytestmark = pytest.mark.gpu
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "safe_denoiser_amd", "csrc")


@pytest.fixture(scope="module")
def alt_lib(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    out = str(tmp_path_factory.mktemp("attn") / "libsdn_attn_notr.so")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function", "-mllvm",
                    "-amdgpu-mfma-vgpr-form", "-DSDN_ATTN_NO_ASM_TR", "-shared", os.path.join(CSRC, "sdn_attn.hip"), "-o", out],
                   check=True, capture_output=True, cwd=CSRC)
    lib = C.CDLL(out)
    for name in ("sdn_attention_bf16", "sdn_attention_f16"):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [C.c_void_p] * 4 + [C.c_int32] * 9 + [C.c_float, C.c_void_p]
    return lib


@pytest.mark.parametrize("d,nq,nk", [(40, 4096, 4096), (40, 1024, 77), (64, 512, 333), (80, 1024, 1024), (80, 300, 77), (160, 256, 256)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_asm_tr_reads_give_the_builtin_bits(alt_lib, d, nq, nk, dt):
    B, H = 2, 8
    g = torch.Generator().manual_seed(d + nq)
    q = torch.randn(B, nq, H * d, generator=g).to(dt).cuda()
    kv = torch.randn(B, nk, 2 * H * d, generator=g).to(dt).cuda()
    k, v = kv[..., :H * d], kv[..., H * d:]
    outs = []
    for lib in (sda.lib(), alt_lib):
        o = torch.full((B, nq, H * d), float("nan"), dtype=dt, device="cuda")
        fn = lib.sdn_attention_bf16 if dt == torch.bfloat16 else lib.sdn_attention_f16
        rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, nq, nk, d, q.stride(1), k.stride(1), v.stride(1), H * d,
                d ** -0.5, _lib.stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
