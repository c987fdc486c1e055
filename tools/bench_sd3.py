#!/usr/bin/env python3
"""BASELINE config 4 as a measurement (not the bench line): SD-v3 medium MMDiT + repellency_methods_fast_sdv3, fp16,
SIDE=128 -> 1024x1024 (or 64 -> 512x512), P prompts per batch, STEPS flow-Euler steps, synthetic weights / references."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.mmdit import SD3Transformer2DModel  # noqa: E402
from safe_denoiser_amd.pipeline_sd3 import SD3SafeDenoiserPipeline  # noqa: E402
from safe_denoiser_amd.repellency import repellency_methods_fast_sdv3 as sd3rep  # noqa: E402
from safe_denoiser_amd.schedulers import FlowMatchEulerDiscreteScheduler  # noqa: E402

SIDE, P, STEPS, M = int(os.environ.get("SIDE", "128")), int(os.environ.get("P", "4")), int(os.environ.get("STEPS", "28")), int(os.environ.get("REFS", "515"))
m = SD3Transformer2DModel(sample_size=SIDE)
m.load_synthetic_on_device(3)
g = torch.Generator(device="cuda").manual_seed(1)
refs = torch.randn(M, 16, SIDE, SIDE, generator=g, device="cuda")
refs = (refs / refs.norm(dim=1, keepdim=True)).cpu()
path = os.path.join(tempfile.mkdtemp(), "proj_ref.pt"); torch.save(refs, path)
proc = sd3rep.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085, 0.012,
                                    n_embed=4, proj_ref_path=path, cache_proj_ref=True, scale=0.03)
emb = torch.randn(2 * P, 333, 4096, device="cuda"); pooled = torch.randn(2 * P, 2048, device="cuda")
pipe = SD3SafeDenoiserPipeline(m, FlowMatchEulerDiscreteScheduler())
run = lambda: pipe(prompt_embeds=emb, pooled_prompt_embeds=pooled, num_inference_steps=STEPS, repellency_processor=proc,
                   generator=[torch.Generator(device="cuda").manual_seed(10 + i) for i in range(P)])
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); out = run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
assert torch.isfinite(out).all()
fl, _ = m.flops(2 * P)
print(f"SD3-medium {SIDE * 8}x{SIDE * 8}, fp16, P={P}, {STEPS} steps, fast_sdv3 repellency M={M}: {P / dt:.3f} images/sec, "
      f"{dt / STEPS * 1e3:.1f} ms per step, window steps {pipe.last_stats['window_steps']}, MMDiT {fl * STEPS / dt / 1e12:.0f} TFLOP/s")
