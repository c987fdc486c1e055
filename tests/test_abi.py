"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/sdn.h declares, the
ctypes table covers them all, argument validation returns SDN_E_INVALID without touching a GPU, and the product
refuses to compute without a GPU (no silent CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

import safe_denoiser_amd as sda
from safe_denoiser_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "sdn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = sda.lib()
    syms = header_symbols()
    assert len(syms) >= 13
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sdn.h but not exported by libsdn.so"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"
    assert lib.sdn_abi_version() >= 1


def test_invalid_arguments_are_rejected_on_host():
    lib = sda.lib()
    assert lib.sdn_cfg_combine(None, 1, 2, 16, 7.5, None, None) == -1
    assert lib.sdn_cfg_combine(0x1000, 1, 4, 16, 7.5, 0x2000, None) == -1          # n_branch must be 2 or 3
    assert lib.sdn_add_noise(0x1000, 0x2000, 6, 1.0, 0.0, 0x3000, None) == -1      # n not a multiple of 4
    assert lib.sdn_add_noise(0x1004, 0x2000, 8, 1.0, 0.0, 0x3000, None) == -1      # misaligned
    assert lib.sdn_pred_x0(0x1000, 0x2000, 8, 0.0, 1.0, 0.0, 0x3000, None) == -1   # sqrt_ac == 0
    p = _lib.RepelParams(n_query=1, n_ref=4, channels=3, hw=3, weight_fn=0, qnorm=0, sigma=1.0)
    assert lib.sdn_repel_apply(C.byref(p), 0x1000, 0x2000, None, None, None, 0x3000, 1 << 20, None) == -1  # D % 4
    p = _lib.RepelParams(n_query=1, n_ref=4, channels=4, hw=4, weight_fn=0, qnorm=0, sigma=0.0)
    assert lib.sdn_repel_apply(C.byref(p), 0x1000, 0x2000, None, None, None, 0x3000, 1 << 20, None) == -1  # sigma
    p = _lib.RepelParams(n_query=1, n_ref=4, channels=4, hw=4, weight_fn=0, qnorm=0, sigma=1.0)
    assert lib.sdn_repel_apply(C.byref(p), 0x1000, 0x2000, None, None, None, 0x3000, 16, None) == -3       # workspace
    assert lib.sdn_repel_workspace_bytes(1, 515, 4, 4096) >= 515 * 16384 * 0 + 2 * 16384 * 4
    assert lib.sdn_repel_workspace_bytes(-1, 1, 1, 1) == 0
    # zero-sized work is a no-op, not an error
    p = _lib.RepelParams(n_query=0, n_ref=4, channels=4, hw=4, weight_fn=0, qnorm=0, sigma=1.0)
    assert lib.sdn_repel_apply(C.byref(p), None, None, None, None, None, None, 0, None) == 0
    assert lib.sdn_add_noise(0x1000, 0x2000, 0, 1.0, 0.0, 0x3000, None) == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_silent_cpu_fallback(tmp_path):
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
    from safe_denoiser_amd.schedulers import DDPMScheduler
    refs = torch.randn(3, 4, 4, 4)
    path = str(tmp_path / "pr.pt")
    torch.save(refs, path)
    proc = thr.get_repellency_method("kernel_fast", torch.zeros(1), None, None, 50, 1000, 0.00085, 0.012, n_embed=4,
                                     proj_ref_path=path, cache_proj_ref=True, beta_threshold=1.0, sigma=3.15)
    with pytest.raises(sda.SdnUnavailable):
        proc.conditioning(torch.randn(1, 4, 4, 4), beta_threshold=True)
    s = DDPMScheduler()
    s.set_timesteps(50)
    with pytest.raises(sda.SdnUnavailable):
        s.add_noise(torch.randn(1, 4, 4, 4), torch.randn(1, 4, 4, 4), 981)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setenv("SDN_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(sda.SdnUnavailable):
        sda.lib()


def test_registry_error_behaviour():
    from safe_denoiser_amd.repellency import (repellency_methods_fast, repellency_methods_fast_sdv3,
                                              repellency_methods_threshold)
    assert set(repellency_methods_threshold.__CONDITIONING_METHOD__) == {"kernel_fast", "sparse"}
    assert {"kernel_fast", "sparse", "random_noise"} <= set(repellency_methods_fast.__CONDITIONING_METHOD__)
    assert {"kernel_fast", "sparse", "random_noise"} <= set(repellency_methods_fast_sdv3.__CONDITIONING_METHOD__)
    with pytest.raises(NameError):
        repellency_methods_threshold.get_repellency_method("lsh", None, None, None, 50, 1000, 0, 0)
    with pytest.raises(NameError):
        repellency_methods_threshold.register_conditioning_method("kernel_fast")(object)
