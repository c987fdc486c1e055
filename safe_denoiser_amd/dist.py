"""Multi-GPU: one process per GPU, prompts sharded across ranks, proj_ref broadcast once.

The reference has no distributed code; its only sharding hook is the manual prompt slice
`--valid_case_numbers start,end` (run_nudity.py:373-375,584).  Prompts are independent units (per-prompt seed,
run_nudity.py:448), so the step loop needs NO collective: rank r takes its shard, rank 0 broadcasts `proj_ref`
(33.75 MB for M=515) and the calibrated `beta_threshold` over RCCL/xGMI at start, counters are gathered at the end.
backend "nccl" is RCCL on ROCm; "gloo" is used for the CPU tests.
"""
from __future__ import annotations

import datetime
import os
import sys
import time
from typing import Optional

import torch
import torch.distributed as dist

# A rank that never arrives must fail the job, not hang it: every collective of this module is an init-time or end-of-run
# exchange, so a generous bound costs nothing (SDN_DIST_TIMEOUT_S overrides).
DEFAULT_TIMEOUT_S = 900


def heartbeat(msg: str, rank: Optional[int] = None):
    """One line on stderr, flushed: `[sdn rank r/W +12.3s] msg`.  Per-rank progress for multi-process runs (a stuck rank is
    then visible in the launcher's log; stdout stays the JSON line)."""
    r = (dist.get_rank() if dist.is_initialized() else int(os.environ.get("RANK", "0"))) if rank is None else rank
    w = dist.get_world_size() if dist.is_initialized() else int(os.environ.get("WORLD_SIZE", "1"))
    print(f"[sdn rank {r}/{w} +{time.perf_counter() - _T0:.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def _single() -> bool:
    """No collective needed: no process group, or a world of one -- unless SDN_DIST_FORCE_COLLECTIVES=1 (tests: RCCL on the one
    GPU a box has runs as a one-rank group, and the point is to execute the collectives)."""
    if not dist.is_initialized():
        return True
    return dist.get_world_size() == 1 and os.environ.get("SDN_DIST_FORCE_COLLECTIVES") != "1"


def check_device_count(world: int, local: int, share: bool = False):
    """Fail fast, before any rendezvous, when this node cannot give every local rank its own GPU."""
    if not torch.cuda.is_available() or share:
        return
    n = torch.cuda.device_count()
    # LOCAL_WORLD_SIZE is torchrun's; a launcher that does not export it (srun / mpirun with RANK + LOCAL_RANK set by hand) may
    # span several nodes, so WORLD_SIZE says nothing about THIS node: then only this rank's own device index can be checked
    lws = os.environ.get("LOCAL_WORLD_SIZE")
    if lws is None:
        # a launcher that exports no LOCAL_WORLD_SIZE: take the scheduler's per-node task count when it states one (srun, mpirun).
        # When NOTHING indicates more than one node (RANK / WORLD_SIZE set by hand on a single machine), WORLD_SIZE is this
        # node's rank count -- so that all ranks fail fast together instead of the low ranks waiting in the rendezvous for the
        # ones that raised.  With several nodes and no per-node count, only this rank's own device index can be checked.
        lws = os.environ.get("SLURM_NTASKS_PER_NODE") or os.environ.get("OMPI_COMM_WORLD_LOCAL_SIZE")

        def _nodes(k):
            v = os.environ.get(k, "")
            return int(v) if v.isdigit() else 1
        if lws is None and max(_nodes(k) for k in ("NNODES", "SLURM_NNODES", "SLURM_JOB_NUM_NODES")) <= 1:
            lws = str(world)
    local_world = int(lws) if lws is not None and str(lws).isdigit() else None
    if (local_world is not None and n < local_world) or local >= n:
        ranks = f"{local_world} ranks on this node" if local_world is not None else "this node's ranks"
        raise RuntimeError(f"{ranks} (LOCAL_RANK {local}) but only {n} GPU(s) visible: one process per "
                           f"GPU is required (set SDN_SHARE_GPU=1 only for a functional rehearsal over gloo)")


def numa_hint(local: int) -> Optional[str]:
    """The `numactl` prefix that binds this rank's host loop next to its GPU, or None when the topology is not exposed.
    Eight host loops (one per GPU) each run a Python thread issuing ~850 launches per forward; on a two-socket MI355X node
    GPUs 0-3 / 4-7 hang off socket 0 / 1, and a rank scheduled on the far socket pays a cross-socket hop per launch and per
    pinned-buffer touch.  The engine never re-execs itself (a process that has touched the GPU must not exec, and the
    launcher hop would change the pid torchrun tracks): bind from the LAUNCHER instead, e.g.
        torchrun --nproc-per-node 8 --no-python bash -c 'exec numactl --cpunodebind=$((LOCAL_RANK/4)) --membind=$((LOCAL_RANK/4)) python bench.py --gpus 8'
    (the exec happens before Python, hence before any HIP call).  This helper only reports the node sysfs names for `local`."""
    try:
        dev = torch.cuda.get_device_properties(local)
        bus = f"{dev.pci_domain_id:04x}:{dev.pci_bus_id:02x}:{dev.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bus}/numa_node") as f:
            node = int(f.read().strip())
        return f"numactl --cpunodebind={node} --membind={node}" if node >= 0 else None
    except Exception:
        return None


def init_from_env(backend: Optional[str] = None, timeout_s: Optional[float] = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not (0 <= rank < world):
        raise RuntimeError(f"RANK {rank} outside WORLD_SIZE {world}")
    # rehearsal on a box with fewer GPUs than ranks (SDN_SHARE_GPU=1): ranks share the devices round-robin and talk over gloo
    # (RCCL refuses two ranks on one device).  Functional check of the N > 1 path only -- never a scaling measurement.
    share = os.environ.get("SDN_SHARE_GPU") == "1" and torch.cuda.is_available()
    if share:
        local = local % max(torch.cuda.device_count(), 1)
        backend = backend or "gloo"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            check_device_count(world, local, share)
            torch.cuda.set_device(local)
        if timeout_s is None:
            timeout_s = float(os.environ.get("SDN_DIST_TIMEOUT_S", DEFAULT_TIMEOUT_S))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def shard_indices(n_items: int, rank: int, world: int, mode: str = "strided"):
    """Prompt indices of this rank.  "strided": r, r+W, ... (balanced tail); "contiguous": the reference's
    valid_case_numbers-style slice [start, end)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    if mode == "strided":
        return list(range(rank, n_items, world))
    if mode == "contiguous":
        per = (n_items + world - 1) // world
        return list(range(min(rank * per, n_items), min((rank + 1) * per, n_items)))
    raise ValueError(mode)


def valid_case_numbers(n_items: int, rank: int, world: int):
    """(start, end) as the reference's --valid_case_numbers would be given to rank r of W."""
    idx = shard_indices(n_items, rank, world, "contiguous")
    return (idx[0], idx[-1] + 1) if idx else (n_items, n_items)


def warm_up_communicator(device) -> float:
    """RCCL builds its communicator (rings over xGMI, IPC handles) lazily inside the FIRST collective; running a 1-element
    all-reduce here keeps that one-off cost (seconds) out of whatever is timed next.  Returns its wall time in ms."""
    if _single():
        return 0.0
    t0 = time.perf_counter()
    t = torch.ones(1, dtype=torch.float32, device=_comm_device(device))
    dist.all_reduce(t)
    if t.is_cuda:
        torch.cuda.synchronize()
    if int(t.item()) != dist.get_world_size():
        raise RuntimeError("communicator warm-up: all_reduce returned a wrong rank count")
    return (time.perf_counter() - t0) * 1e3


def _comm_device(device):
    """Tensors of a collective live on the GPU for RCCL and on the CPU for gloo (tests, shared-GPU rehearsals)."""
    return torch.device("cpu") if dist.is_initialized() and dist.get_backend() == "gloo" else device


def broadcast_proj_ref(refs: Optional[torch.Tensor], device, src: int = 0) -> torch.Tensor:
    """Rank `src` holds refs [M,C,H,W] fp32; everyone returns the same tensor on `device`.  Verifies the payload
    with an fp64 checksum all-reduced MIN/MAX (cheap, once per run)."""
    if _single():
        return refs.to(device)
    rank = dist.get_rank()
    cdev = _comm_device(device)
    meta = torch.zeros(4, dtype=torch.int64, device=cdev)
    if rank == src:
        meta = torch.tensor(list(refs.shape), dtype=torch.int64, device=cdev)
    dist.broadcast(meta, src=src)
    buf = refs.to(device=cdev, dtype=torch.float32).contiguous() if rank == src else \
        torch.empty(tuple(meta.tolist()), dtype=torch.float32, device=cdev)
    dist.broadcast(buf, src=src)
    chk = buf.double().sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if float(lo) != float(hi):
        raise RuntimeError("proj_ref broadcast checksum mismatch across ranks")
    return buf.to(device)


def broadcast_scalar(value: float, device, src: int = 0) -> float:
    if _single():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_comm_device(device))
    dist.broadcast(t, src=src)
    return float(t.item())


def barrier():
    if not _single():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if _single():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device) -> float:
    if _single():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value: float, device) -> list:
    """Every rank's `value`, in rank order, on every rank (end-of-run per-rank counters; never inside the step loop)."""
    if _single():
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=_comm_device(device))
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def throughput_over_ranks(units_mine: int, dt_mine: float, dt_window_mine: float, device) -> dict:
    """Whole-job throughput of a prompt-sharded run (bench.py's `value`): every rank reports the units it processed, its own
    elapsed time and its view of the common timed window (barrier ... barrier).  value = all units / the LONGEST window
    (MAX over ranks); per_rank = each rank's own units / its own time (N entries, rank order)."""
    units = gather_over_ranks(float(units_mine), device)
    times = gather_over_ranks(float(dt_mine), device)
    window = max_over_ranks(float(dt_window_mine), device)
    return {"value": sum(units) / window, "window_s": window, "units": [int(u) for u in units],
            "per_rank": [u / t for u, t in zip(units, times)]}
