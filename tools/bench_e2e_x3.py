"""bench.py's `e2e_bf16x3` leg alone (the headline call in the tolerance-meeting mode), for iteration on the bf16x3 path:
    python tools/bench_e2e_x3.py [P=32] [timed_batches=1] [inference_steps=50]"""
import argparse
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from safe_denoiser_amd.repellency import repellency_methods_threshold as thr  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda", 0)
args = argparse.Namespace(dtype="bf16", scheduler="ddpm", inference_steps=steps)
g = torch.Generator().manual_seed(0)
refs = torch.randn(515, 4, 64, 64, generator=g)
refs = refs / refs.norm(dim=1, keepdim=True)
path = os.path.join(tempfile.mkdtemp(prefix="sdn_x3_"), "pr.pt")
torch.save(refs, path)
proc = thr.get_repellency_method("kernel_fast", torch.zeros(1, device=dev), None, None, 50, 1000, 0.00085, 0.012, n_embed=16,
                                 scale=0.33, sigma=3.15, beta_threshold_margin=1.6, proj_ref_path=path, cache_proj_ref=True,
                                 beta_threshold=1.6 + 5e-9)        # every (prompt, window step) pair fires, as in bench.py on synthetic weights
res = bench.measure_e2e_precision(args, dev, proc, P, list(range(515)), timed_batches=nb)
print(res)
