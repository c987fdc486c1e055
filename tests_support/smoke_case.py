"""One small invocation of the hot path on cuda:0, checked against the CPU oracle."""
import tempfile

import torch

from oracle import repellency as orp


def run_smoke():
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
    g = torch.Generator().manual_seed(0)
    refs = orp.channel_normalise(torch.randn(33, 4, 16, 16, generator=g))
    x = torch.randn(2, 4, 16, 16, generator=g)
    with tempfile.TemporaryDirectory() as td:
        path = f"{td}/pr.pt"
        torch.save(refs, path)
        proc = thr.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085,
                                         0.012, n_embed=4, proj_ref_path=path, cache_proj_ref=True, sigma=3.15,
                                         scale=0.33, beta_threshold=2.0, beta_threshold_margin=1.6)
    xg = x.clone().cuda()
    _neg, den, isneg = proc.conditioning_device(xg, beta_threshold=True)
    for i in range(2):
        xi = x[i:i + 1].clone()
        o = orp.kernel_fast_conditioning(xi, refs, flavour="threshold", scale=0.33, sigma=3.15, beta_threshold=2.0,
                                         beta_threshold_margin=1.6, use_beta_threshold=True)
        torch.testing.assert_close(xg[i:i + 1].cpu(), o["x_0_hat"], rtol=2e-5, atol=2e-6)
        assert bool(isneg[i].item()) == o["is_negation"]
