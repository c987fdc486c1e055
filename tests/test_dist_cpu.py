"""N > 1 path on CPU: world_size-2 gloo processes exercise the prompt sharding, the proj_ref / threshold
broadcast with checksum, and the end-of-run reductions bench.py relies on."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from safe_denoiser_amd import dist as sdist


def test_shards_partition_every_prompt_exactly_once():
    for n in (0, 1, 7, 515, 994, 10000):
        for w in (1, 2, 4, 8):
            for mode in ("strided", "contiguous"):
                got = sorted(i for r in range(w) for i in sdist.shard_indices(n, r, w, mode))
                assert got == list(range(n)), (n, w, mode)
            sizes = [len(sdist.shard_indices(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
    assert sdist.valid_case_numbers(515, 1, 8) == (65, 130)           # run_nudity.py --valid_case_numbers 65,130
    assert sdist.valid_case_numbers(3, 7, 8) == (3, 3)
    with pytest.raises(ValueError):
        sdist.shard_indices(5, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = sdist.init_from_env(backend="gloo")
    dev = torch.device("cpu")
    refs = None
    if r == 0:
        g = torch.Generator().manual_seed(0)
        refs = torch.randn(9, 4, 8, 8, generator=g)
        refs = refs / refs.norm(dim=1, keepdim=True)
    got = sdist.broadcast_proj_ref(refs, dev)
    thr = sdist.broadcast_scalar(3.25 if r == 0 else -1.0, dev)
    mine = sdist.shard_indices(11, r, w)
    sdist.barrier()
    slowest = sdist.max_over_ranks(1.0 + r, dev)
    total = sdist.sum_over_ranks(len(mine), dev)
    warm = sdist.warm_up_communicator(dev)
    sdist.heartbeat("worker done")
    # rank r processed 10 (r + 1) images in (1 + r) s of its own clock; the common window is the slower rank's
    tp = sdist.throughput_over_ranks(10 * (r + 1), 1.0 + r, 1.0 + r, dev)
    q.put((r, float(got.double().sum()), tuple(got.shape), thr, mine, slowest, total, warm >= 0.0, tp))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, sh0, t0, m0, sl0, tot0, w0, tp0), (r1, s1, sh1, t1, m1, sl1, tot1, w1, tp1) = res
    assert w0 and w1
    # bench.py's aggregation: N per-rank entries; value = all units over the common (longest) window
    assert tp0 == tp1 and tp0["per_rank"] == [10.0, 10.0] and tp0["units"] == [10, 20] and tp0["window_s"] == 2.0
    assert tp0["value"] == 15.0 == sum(tp0["units"]) / tp0["window_s"]
    assert s0 == s1 and sh0 == sh1 == (9, 4, 8, 8)                 # same proj_ref everywhere
    assert t0 == t1 == 3.25                                        # every rank gates identically
    assert m0 == [0, 2, 4, 6, 8, 10] and m1 == [1, 3, 5, 7, 9]
    assert sl0 == sl1 == 2.0 and tot0 == tot1 == 11.0


def test_init_rejects_bad_rank_and_reports_missing_gpus(monkeypatch):
    for k in ("LOCAL_WORLD_SIZE", "SLURM_NNODES", "SLURM_JOB_NUM_NODES", "NNODES", "SLURM_NTASKS_PER_NODE", "OMPI_COMM_WORLD_LOCAL_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "5")
    with pytest.raises(RuntimeError):
        sdist.init_from_env(backend="gloo")
    # one process per GPU: 4 local ranks on a node that shows fewer devices must fail before any rendezvous
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    with pytest.raises(RuntimeError, match="only 2 GPU"):
        sdist.check_device_count(4, 3)
    sdist.check_device_count(4, 3, share=True)                      # the gloo rehearsal mode is allowed to share devices
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    sdist.check_device_count(2, 1)
    # a launcher that exports no LOCAL_WORLD_SIZE, 16 ranks on 2 x 8 GPUs: WORLD_SIZE is not this node's rank count.  srun says
    # how many nodes there are: only the rank's own device index is checked ...
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setenv("SLURM_NNODES", "2")
    sdist.check_device_count(16, 7)
    with pytest.raises(RuntimeError, match="only 8 GPU"):
        sdist.check_device_count(16, 8)
    # ... or the scheduler states the per-node count itself (mpirun)
    monkeypatch.delenv("SLURM_NNODES")
    monkeypatch.setenv("OMPI_COMM_WORLD_LOCAL_SIZE", "8")
    sdist.check_device_count(16, 7)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    with pytest.raises(RuntimeError, match="8 ranks on this node"):
        sdist.check_device_count(16, 2)
    # nothing indicates several nodes (RANK / WORLD_SIZE set by hand on one machine with 4 GPUs): WORLD_SIZE is the node's rank
    # count, so EVERY rank refuses -- not only ranks 4-7 while 0-3 wait in the rendezvous
    monkeypatch.delenv("OMPI_COMM_WORLD_LOCAL_SIZE")
    for r_ in (0, 3, 7):
        with pytest.raises(RuntimeError, match="only 4 GPU"):
            sdist.check_device_count(8, r_)


class _Img:
    def save(self, path):
        open(path, "wb").write(b"png")


class _Pipe:
    variant = "threshold_time"

    def __call__(self, prompts, **kw):
        assert len(kw["generator"]) == len(prompts) == len(kw["guidance_scale"])
        return [_Img() for _ in prompts]


def _job_worker(rank, world, port, root, q):
    """driver.run_job on two gloo ranks: each rank walks its `rank::world` shard of one prompt table into its own tree."""
    import json
    from safe_denoiser_amd import driver
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = sdist.init_from_env(backend="gloo")
    args = driver.parse_args(["--config", os.path.join(root, "cfg.json")])
    verdict = lambda imgs, threshold: (True, 0.75)
    art = driver.run_job(args, _Pipe(), None, None, eval_func=verdict, prompts_per_batch=4, rank=r, world=w, device="cpu")
    sdist.barrier()                                                    # every tree is complete before anyone merges
    n_all = sdist.sum_over_ranks(art.safe_cnt + art.unsafe_cnt, torch.device("cpu"))
    merged = driver.merge_rank_outputs(args.save_dir, w) if r == 0 else None
    q.put((r, sorted(os.listdir(os.path.join(art.save_dir, "all"))), n_all, merged))
    dist.destroy_process_group()


def test_run_job_on_two_gloo_ranks_writes_every_case_exactly_once(tmp_path):
    """run_nudity.py:373-375,582-584 (one process per GPU, each on its slice) as the engine runs it: per-rank trees whose union
    is the reference's single tree, and the merged detect_dict."""
    import json
    n = 11
    rows = ["case_number,prompt,categories,evaluation_seed,guidance"] + [f'{100 + i},"p {i}",sexual,{i},{7.5 if i % 2 else 9.0}' for i in range(n)]
    (tmp_path / "p.csv").write_text("\n".join(rows) + "\n")
    (tmp_path / "cfg.json").write_text(json.dumps({"erase_id": "safree_neg_prompt_rep_threshold_time", "nudity": "nudity",
                                                   "data": str(tmp_path / "p.csv"), "save_dir": str(tmp_path / "out")}))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, files0, n0, merged), (_, files1, n1, _none) = res
    assert n0 == n1 == float(n)
    assert files0 == sorted(f"{100 + i}_sexual.png" for i in range(0, n, 2)) and files1 == sorted(f"{100 + i}_sexual.png" for i in range(1, n, 2))
    assert sorted(os.listdir(tmp_path / "out")) == ["detect_dict.json", "rank00", "rank01"]
    assert len(merged["unsafe"]) == n and merged["toxic_size"] == {"sexual": n, "average": n} and merged["toxic_ratio"]["average"] == 1.0
