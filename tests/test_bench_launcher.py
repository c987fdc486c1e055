"""`python bench.py --gpus N` must BE N ranks: the parent starts N child processes with the torchrun
environment contract and relays rank 0's JSON; a mismatching WORLD_SIZE is an error, not a warning.  CPU rehearsal
(`--launch-check`: gloo, no GPU call)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_gpus_2_starts_two_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rank_sum"] == 1.0          # ranks 0 and 1 both joined the process group


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--launch-check"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0
    assert "WORLD_SIZE" in r.stderr


def test_failing_rank_fails_the_launch():
    r = _run(["--gpus", "2", "--launch-check", "--no-such-flag"])
    assert r.returncode != 0
