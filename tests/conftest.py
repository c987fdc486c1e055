import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the GPU suite runs under a wall-clock limit: keep the slowest tests visible in every log
    if getattr(config.option, "durations", None) is None:
        config.option.durations = 15
        config.option.durations_min = 1.0


# GPU files in the order they should run: every oracle / golden PARITY file first (an overrun of the suite's limit must never cut
# parity), then properties, A/B equalities, launch / driver / multi-process plumbing.  Files not listed keep their place after these.
_GPU_ORDER = ["test_gpu_repellency", "test_gpu_schedulers", "test_gpu_rng", "test_gpu_ops", "test_gpu_x3t", "test_gpu_unet", "test_gpu_clip",
              "test_gpu_vae", "test_gpu_pipeline", "test_gpu_safree_call", "test_gpu_f32", "test_gpu_e2e_ids", "test_gpu_mmdit",
              "test_gpu_properties", "test_gpu_driver", "test_gpu_from_pretrained", "test_gpu_attn_asm_equality", "test_gpu_rccl_smoke",
              "test_gpu_bench_two_ranks"]


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(_GPU_ORDER)}

    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return rank.get(mod, len(rank))
    items.sort(key=key)                       # stable: the order inside a file is unchanged


@pytest.fixture(scope="session")
def sd14_full_state_dict():
    """The synthetic full-size SD-v1.4 state_dict (859.5 M parameters, seed 1234) the full-size parity tests share: generating it
    takes ~10 s of host time per test otherwise."""
    from safe_denoiser_amd.unet import UNet2DConditionModel
    return UNet2DConditionModel(text_len=77).synthetic_state_dict(1234)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "repellency_golden.npz")
    z = np.load(path)
    cases = {}
    for key in z.files:
        if key == "__cases__":
            continue
        name, field = key.split("/", 1)
        cases.setdefault(name, {})[field] = z[key]
    return cases
