import sys, time, torch
sys.path.insert(0, "/root/repo")
from safe_denoiser_amd.unet import UNet2DConditionModel
u = UNet2DConditionModel(latent_repeat=2); u.load_synthetic_on_device(1)
for P in (1, 4):
    x = torch.randn(P, 4, 64, 64, device="cuda"); e = u.prepare_text(torch.randn(2 * P, 77, 768, device="cuda")); out = torch.empty(2 * P, 4, 64, 64, device="cuda")
    for mode, sk in ((False, False), (True, False), (True, True)):
        u.set_graph_mode(mode); u.set_split_k(sk)
        for _ in range(3): u.forward_into(x, 500.0, e, out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20): u.forward_into(x, 500.0 - i, e, out)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"P={P} graph={mode} split_k={sk}: host issue {1e3*(t1-t0)/20:.2f} ms/forward, total {1e3*(t2-t0)/20:.2f} ms/forward")
