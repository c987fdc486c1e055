"""Host-side sanitizer run (SURVEY section 5 "race detection / sanitizers"): the launch-plan builder, arena allocator and
parameter manifest (csrc/sdn_unet.hip, host code) rebuilt with -fsanitize=address,undefined (`make asan`) and driven by the
host-logic tests -- UNet / MMDiT / VAE / CLIP plan creation, manifests, workspace sizing, FLOP counts, argument validation of
every entry point -- in a child process with the clang ASan runtime preloaded.  Any finding aborts the child."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_planner_under_asan_ubsan(tmp_path):
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        pytest.skip("clang ASan runtime not found")
    out = str(tmp_path / "asan")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "safe_denoiser_amd", "csrc"), "asan", f"ASAN_OUT={out}", "ARCH=gfx950"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, SDN_LIB=os.path.join(out, "libsdn_asan.so"), LD_PRELOAD=rt[-1],
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_unet_host.py",
                        "tests/test_abi.py", "tests/test_oracle_vae.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
