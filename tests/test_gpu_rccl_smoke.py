"""RCCL on the GPU box: the collectives of safe_denoiser_amd.dist (broadcast of proj_ref + checksum all-reduces, scalar broadcast,
max / sum / gather over ranks, barrier) executed on the `nccl` backend -- which IS RCCL on ROCm -- with the one GPU a box has:
a ONE-rank process group (RCCL refuses two ranks on one device, and no multi-GPU node is available to the builder).  It does not
measure xGMI; it shows that the library loads under this image's environment (HSA_ENABLE_IPC_MODE_LEGACY=0), builds a communicator
and runs every collective the 8-GPU launch issues, on device tensors, through the same helper functions -- their world-size-1
early returns switched off (SDN_DIST_FORCE_COLLECTIVES=1)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SDN_ROOT"])
from safe_denoiser_amd import dist as sdist
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["SDN_PORT"])
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
assert dist.get_backend() == "nccl"
g = torch.Generator().manual_seed(0)
refs = torch.randn(515, 4, 64, 64, generator=g)
refs = refs / refs.norm(dim=1, keepdim=True)
ms = sdist.warm_up_communicator(dev)
got = sdist.broadcast_proj_ref(refs, dev)
assert got.is_cuda and torch.equal(got.cpu(), refs)
assert sdist.broadcast_scalar(3.25, dev) == 3.25
assert sdist.max_over_ranks(2.5, dev) == 2.5 and sdist.sum_over_ranks(7.0, dev) == 7.0 and sdist.gather_over_ranks(1.5, dev) == [1.5]
tp = sdist.throughput_over_ranks(128, 16.0, 16.5, dev)
assert tp["value"] == 128 / 16.5 and tp["per_rank"] == [8.0]
sdist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK", round(ms, 1))
"""


def test_rccl_runs_every_collective_of_the_n_rank_launch_on_one_rank(tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SDN_ROOT=ROOT, SDN_PORT=str(port), SDN_DIST_FORCE_COLLECTIVES="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    path = tmp_path / "rccl_one_rank.py"
    path.write_text(SCRIPT)
    r = subprocess.run([sys.executable, str(path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
