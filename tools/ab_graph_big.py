"""Does replaying the forward as one hipGraph help at the benchmark batch too?  (it exists for the launch-bound small batches)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.unet import UNet2DConditionModel
B = int(os.environ.get("B", "192"))
u = UNet2DConditionModel(latent_repeat=3); u.load_synthetic_on_device(1234)
x = torch.randn(B // 3, 4, 64, 64, device="cuda"); tb = u.prepare_text(torch.randn(B, 77, 768, device="cuda")); y = torch.empty(B, 4, 64, 64, device="cuda")
for rnd in range(3):
    for mode in (False, True):
        u.set_graph_mode(mode)
        for _ in range(3):
            u.forward_into(x, 981.0, tb, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            u.forward_into(x, 981.0, tb, y)
        torch.cuda.synchronize()
        print(f"round {rnd} graph={mode}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per forward", flush=True)
