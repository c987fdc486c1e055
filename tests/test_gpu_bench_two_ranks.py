"""The N > 1 path of bench.py on the GPU box's ONE card: `python bench.py --gpus 2` with SDN_SHARE_GPU=1 starts two ranks that
share the device and talk over gloo (RCCL refuses two ranks on one device) -- a FUNCTIONAL run of everything the 8-GPU launch does
except the xGMI transport: the self-launcher, the rendezvous, the prompt shard r::2, the proj_ref broadcast + checksum, the gate
broadcast, each rank's end-to-end batches, the throughput aggregation over ranks.  Not a scaling measurement (the line says so).
Small shapes of the real workload: 4 prompts per batch, a 6-step schedule (t = 831 is inside the repellency window), full SD-v1.4
UNet / CLIP / VAE."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_the_gpu_and_produce_one_aggregated_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env.update(SDN_SHARE_GPU="1", SDN_DIST_TIMEOUT_S="300")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--prompts-per-batch", "4", "--inference-steps", "6", "--total-prompts", "9", "--no-extras", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["unit"] == "images/sec"
    cfg = line["config"]
    assert len(cfg["per_rank_images_per_sec"]) == 2 and all(v > 0 for v in cfg["per_rank_images_per_sec"])
    assert cfg["images_timed"] == 8 and cfg["parallelism"] == "prompt-shard x2" and "rehearsal" in cfg
    assert cfg["proj_ref_broadcast_ms"] is not None and cfg["proj_ref_broadcast_ms"] > 0          # the broadcast really happened
    assert cfg["renoise_draws_rank0"] > 0                                                          # window steps fired on rank 0
    # value = all images over the longest rank window: never above the sum of the per-rank rates
    assert 0 < line["value"] <= sum(cfg["per_rank_images_per_sec"]) * 1.001
    assert "[sdn rank 1/2" in r.stderr                                                             # rank 1's heartbeats reached the launcher's log
