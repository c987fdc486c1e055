"""`SafeDenoiserPipeline.from_pretrained(local_dir)` + the arguments of the reference's call site (run_nudity.py:104-131,439-460): a
synthetic diffusers-layout directory written by the test (small UNet / VAE / CLIP configs, safetensors weights, the SD-v1.4
scheduler file) -> PIL images, identical to the pipeline built directly from the same state dicts."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import repellency as orp
from safe_denoiser_amd import driver
from safe_denoiser_amd.clip import CLIPTextModel
from safe_denoiser_amd.pipeline import SafeDenoiserPipeline
from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
from safe_denoiser_amd.schedulers import DDPMScheduler
from safe_denoiser_amd.unet import UNet2DConditionModel
from safe_denoiser_amd.vae import AutoencoderKL
from tests.test_checkpoint import SD14_SCHEDULER_JSON
from tests.test_gpu_pipeline import SMALL
from tests.test_gpu_safree_call import CLIP_CFG
from tests_support.fake_tokenizer import FakeCLIPTokenizer

pytestmark = pytest.mark.gpu
VAE_CFG = dict(block_out_channels=(64, 128), layers_per_block=1, sample_size=32)


def _write_checkpoint(root):
    from safetensors.torch import save_file
    u = UNet2DConditionModel(text_len=77, **SMALL)
    usd = u.synthetic_state_dict(11)
    v = AutoencoderKL(**VAE_CFG)
    vsd = v.synthetic_state_dict(5)
    c = CLIPTextModel(dtype=torch.float16, **CLIP_CFG)
    csd = c.synthetic_state_dict(31)
    for sub, cfg, sd, fname in (
            ("unet", dict(SMALL, up_block_types=["UpBlock2D", "CrossAttnUpBlock2D"], act_fn="silu", in_channels=4, out_channels=4,
                          norm_num_groups=32, _class_name="UNet2DConditionModel"), usd, "diffusion_pytorch_model.safetensors"),
            ("vae", dict(VAE_CFG, act_fn="silu", latent_channels=4, scaling_factor=0.18215, _class_name="AutoencoderKL"), vsd,
             "diffusion_pytorch_model.safetensors"),
            ("text_encoder", dict(CLIP_CFG, hidden_act="quick_gelu", _class_name="CLIPTextModel"),
             {"text_model." + k: t for k, t in csd.items()}, "model.safetensors")):
        os.makedirs(os.path.join(root, sub))
        json.dump({k: (list(x) if isinstance(x, tuple) else x) for k, x in cfg.items()}, open(os.path.join(root, sub, "config.json"), "w"))
        save_file({k: t.contiguous() for k, t in sd.items()}, os.path.join(root, sub, fname))
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump(SD14_SCHEDULER_JSON, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    return usd, vsd, csd


def test_from_pretrained_then_the_reference_call_site_returns_pil_images(tmp_path):
    root = str(tmp_path / "ckpt")
    usd, vsd, csd = _write_checkpoint(root)
    tok = FakeCLIPTokenizer(vocab_size=CLIP_CFG["vocab_size"])
    # load_sd (run_nudity.py:104-131)
    scheduler = DDPMScheduler.from_pretrained(root, subfolder="scheduler")
    pipe = SafeDenoiserPipeline.from_pretrained(root, scheduler=scheduler, torch_dtype=torch.bfloat16, revision="fp16",
                                                variant=driver.ERASE_IDS["safree_neg_prompt_rep_threshold_time"][1],
                                                latent_repeat=3, tokenizer=tok)
    pipe = pipe.to("cuda:0")
    assert pipe.scheduler.config.clip_sample is False and pipe.vae is not None and pipe.text_encoder is not None
    refs = orp.channel_normalise(torch.randn(12, 4, 16, 16, generator=torch.Generator().manual_seed(4)))
    torch.save(refs, tmp_path / "proj_ref.pt")
    proc = thr.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 6, 1000, 0.00085, 0.012, n_embed=4,
                                     scale=0.33, sigma=3.15, proj_ref_path=str(tmp_path / "proj_ref.pt"), cache_proj_ref=True,
                                     beta_threshold=1e-6, beta_threshold_margin=1e9)

    # the arguments the reference's call site hands over (run_nudity.py:439-460), by name, built in this test's own words:
    # README-default flags of configs/base/vanilla/safree_neg_prompt_config.json (safree + svf + lra on, re_attn_t "-1,1001")
    seed = 2868251644
    negative_prompt_space = driver.NUDITY_NEGATIVE_PROMPT_SPACE
    gen = torch.Generator(device="cuda")
    flags = dict(safree=True, svf=True, lra=True, alpha=0.01, up_t=10, category="nudity", logger=None, re_attn_t=[-1, 1001])
    call_kwargs = dict(num_images_per_prompt=1, guidance_scale=7.5, num_inference_steps=6, height=128, width=128,
                       negative_prompt=", ".join(negative_prompt_space), negative_prompt_space=negative_prompt_space,
                       repellency_processor=proc, safree_dict=flags)
    target_prompt = "a painting of a woman standing near the sea , lustful mood"

    def call(pipe):
        return pipe(target_prompt, generator=gen.manual_seed(seed), **call_kwargs)

    imgs = call(pipe)
    assert isinstance(imgs, list) and len(imgs) == 1 and imgs[0].size == (32, 32) and imgs[0].mode == "RGB"
    assert pipe.last_stats["branches"] == 3 and pipe.last_stats["renoise_draws"] > 0
    # the same stack built by hand from the same state dicts gives the same image, bit for bit
    u = UNet2DConditionModel(text_len=77, latent_repeat=3, **SMALL); u.load_state_dict(usd)
    v = AutoencoderKL(**VAE_CFG); v.load_state_dict(vsd)
    assert pipe.text_encoder.precision == "bf16x3"                   # the default in every mode: its output feeds categorical decisions
    c = CLIPTextModel(precision="bf16x3", **CLIP_CFG); c.load_state_dict(csd)
    direct = SafeDenoiserPipeline(u, DDPMScheduler(), variant="threshold_time", vae=v, text_encoder=c, tokenizer=tok)
    assert np.array_equal(np.asarray(call(direct)[0]), np.asarray(imgs[0]))
    legacy = SafeDenoiserPipeline.from_pretrained(root, scheduler=scheduler, torch_dtype=torch.bfloat16, latent_repeat=3, tokenizer=tok,
                                                  text_encoder_precision=None)
    assert legacy.text_encoder.precision is None and legacy.text_encoder.dtype == torch.bfloat16
    # what the call surface rejects instead of silently dropping
    with pytest.raises(NotImplementedError):
        pipe(target_prompt, num_images_per_prompt=2, num_inference_steps=2)
    with pytest.raises(TypeError):
        pipe(target_prompt, num_inference_steps=2, not_an_argument=1)
    seen = []
    out = pipe(target_prompt, num_inference_steps=3, callback=lambda i, t, lat: seen.append((i, int(t), tuple(lat.shape))),
               callback_steps=2, output_type="np", return_dict=False, safree_dict={"lra": True})
    assert seen == [(0, 667, (1, 4, 16, 16)), (2, 1, (1, 4, 16, 16))] and out.shape == (1, 32, 32, 3)


def test_from_pretrained_scheduled_precision_builds_both_plans(tmp_path):
    """`from_pretrained(dir, precision="scheduled")`: the fp16 plan + the bf16x3 plan over the same checkpoint, bf16x3 inside the
    repellency window, text encoder bf16x3 -- the same images as the stack built by hand."""
    root = str(tmp_path / "ckpt")
    usd, vsd, csd = _write_checkpoint(root)
    tok = FakeCLIPTokenizer(vocab_size=CLIP_CFG["vocab_size"])
    pipe = SafeDenoiserPipeline.from_pretrained(root, precision="scheduled", variant="threshold_time", latent_repeat=3, tokenizer=tok)
    assert pipe.unet.dtype == torch.float16 and pipe.unet_hi.precision == "bf16x3" and pipe.text_encoder.precision == "bf16x3"
    assert pipe.precision_schedule == {"window": True}
    refs = orp.channel_normalise(torch.randn(12, 4, 16, 16, generator=torch.Generator().manual_seed(4)))
    torch.save(refs, tmp_path / "proj_ref.pt")
    proc = thr.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 6, 1000, 0.00085, 0.012, n_embed=4,
                                     scale=0.33, sigma=3.15, proj_ref_path=str(tmp_path / "proj_ref.pt"), cache_proj_ref=True,
                                     beta_threshold=1e-6, beta_threshold_margin=1e9)
    space = driver.NUDITY_NEGATIVE_PROMPT_SPACE
    kw = dict(guidance_scale=7.5, num_inference_steps=10, height=128, width=128, negative_prompt=", ".join(space), negative_prompt_space=space,
              repellency_processor=proc, safree_dict=dict(safree=True, svf=True, lra=True, alpha=0.01, up_t=10, category="nudity", logger=None,
                                                          re_attn_t=[-1, 1001]))
    prompts = ["a painting of a woman standing near the sea , lustful mood", "two cats asleep on a red sofa"]
    gens = lambda: [torch.Generator(device="cuda").manual_seed(7 + i) for i in range(2)]
    imgs = pipe(prompts, generator=gens(), **kw)
    assert pipe.last_stats["hi_steps"] == pipe.last_stats["window_steps"] == 2 and pipe.last_stats["window_readbacks"] == 0
    lo = UNet2DConditionModel(text_len=77, dtype=torch.float16, latent_repeat=3, **SMALL); lo.load_state_dict(usd)
    hi = UNet2DConditionModel(text_len=77, precision="bf16x3", latent_repeat=3, **SMALL); hi.load_state_dict(usd)
    v = AutoencoderKL(**VAE_CFG); v.load_state_dict(vsd)
    c = CLIPTextModel(precision="bf16x3", **CLIP_CFG); c.load_state_dict(csd)
    direct = SafeDenoiserPipeline(lo, DDPMScheduler(), variant="threshold_time", vae=v, text_encoder=c, tokenizer=tok, unet_hi=hi,
                                  precision_schedule={"window": True})
    for a, b in zip(direct(prompts, generator=gens(), **kw), imgs):
        assert np.array_equal(np.asarray(a), np.asarray(b))
