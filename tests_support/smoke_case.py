"""One small invocation of the hot path on cuda:0, checked against the CPU oracle: (1) a batched repellency projection,
(2) a 3-step denoising loop (small UNet + CFG + DDPM + repellency window) for one prompt on a fixed noise tape."""
import tempfile

import torch

from oracle import pipeline as opipe
from oracle import repellency as orp
from oracle import schedulers as osch
from oracle.unet import OracleUNet


def _proc(refs, **params):
    from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
    with tempfile.TemporaryDirectory() as td:
        path = f"{td}/pr.pt"
        torch.save(refs, path)
        return thr.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085,
                                         0.012, n_embed=4, proj_ref_path=path, cache_proj_ref=True, **params)


def run_smoke():
    g = torch.Generator().manual_seed(0)
    # ---- (1) projection --------------------------------------------------------------------------------
    refs = orp.channel_normalise(torch.randn(33, 4, 16, 16, generator=g))
    x = torch.randn(2, 4, 16, 16, generator=g)
    params = dict(sigma=3.15, scale=0.33, beta_threshold=2.0, beta_threshold_margin=1.6)
    proc = _proc(refs, **params)
    xg = x.clone().cuda()
    _neg, _den, isneg = proc.conditioning_device(xg, beta_threshold=True)
    for i in range(2):
        o = orp.kernel_fast_conditioning(x[i:i + 1].clone(), refs, flavour="threshold", use_beta_threshold=True, **params)
        torch.testing.assert_close(xg[i:i + 1].cpu(), o["x_0_hat"], rtol=2e-5, atol=2e-6)
        assert bool(isneg[i].item()) == o["is_negation"]
    # ---- (2) the loop ----------------------------------------------------------------------------------
    from safe_denoiser_amd.pipeline import SafeDenoiserPipeline
    from safe_denoiser_amd.schedulers import DDPMScheduler
    from safe_denoiser_amd.unet import UNet2DConditionModel
    cfg = dict(block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
               layers_per_block=1, attention_head_dim=8, cross_attention_dim=768, sample_size=16)
    ocfg = dict(block_out_channels=(320, 640), level_has_attn=(True, False), layers_per_block=1, n_heads=8,
                cross_dim=768, sample_size=16)
    u = UNet2DConditionModel(text_len=77, dtype=torch.float16, **cfg)
    sd = u.synthetic_state_dict(3)
    u.load_state_dict(sd)
    E = torch.randn(2, 77, 768, generator=g)
    tape = torch.randn(16, 1, 4, 16, 16, generator=g)

    class Tape:
        def __init__(self):
            self.i = 0

        def __call__(self, p, shape):
            z = tape[self.i].reshape(shape).clone()
            self.i += 1
            return z

    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)     # gate always fires
    ref, st = opipe.denoise_one(OracleUNet(sd, ocfg, act_dtype=torch.float16), osch.DDPM(), E, 0, Tape(),
                                num_inference_steps=3, repel=dict(flavour="threshold", proj_refs=refs, **params),
                                negation_warmup_end=0)
    tp = Tape()
    pipe = SafeDenoiserPipeline(u, DDPMScheduler())
    lat = pipe(prompt_embeddings=E.cuda(), num_inference_steps=3, repellency_processor=_proc(refs, **params), noise_fn=tp,
               negation_warmup_end=0, return_latents=True)
    torch.cuda.synchronize()
    rel = float((lat.cpu() - ref).norm() / ref.norm())
    assert pipe.last_stats["renoise_draws"] == st["renoise_draws"] > 0, (pipe.last_stats, st)
    assert rel <= 1e-2, rel
    print(f"smoke: projection ok; 3-step loop rel L2 vs oracle {rel:.2e}, re-noise draws {st['renoise_draws']}")
