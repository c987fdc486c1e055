"""Checkpoint-directory loader, host side: scheduler_config.json semantics of
`DDPMScheduler.from_pretrained(model_id, subfolder="scheduler")` (run_nudity.py:108), weight-file discovery, config mapping."""
import json
import os

import pytest
import torch

from safe_denoiser_amd import checkpoint as ck
from safe_denoiser_amd.schedulers import DDIMScheduler, DDPMScheduler

# the scheduler file SD-v1.4 ships (PNDM-authored; the keys DDPM does not accept are ignored by from_config)
SD14_SCHEDULER_JSON = {"_class_name": "PNDMScheduler", "_diffusers_version": "0.7.0.dev0", "beta_end": 0.012,
                       "beta_schedule": "scaled_linear", "beta_start": 0.00085, "num_train_timesteps": 1000,
                       "set_alpha_to_one": False, "skip_prk_steps": True, "steps_offset": 1, "trained_betas": None,
                       "clip_sample": False}


def _write(tmp_path, cfg):
    d = tmp_path / "scheduler"
    d.mkdir(parents=True, exist_ok=True)
    (d / "scheduler_config.json").write_text(json.dumps(cfg))
    return str(tmp_path)


def test_scheduler_from_pretrained_follows_the_file(tmp_path):
    s = DDPMScheduler.from_pretrained(_write(tmp_path, SD14_SCHEDULER_JSON), subfolder="scheduler")
    assert s.config.clip_sample is False and s.config.steps_offset == 1 and s.config.beta_schedule == "scaled_linear"
    direct = DDPMScheduler()                                             # the engine's SD-v1.4 defaults = that file
    assert torch.equal(s.alphas_cumprod, direct.alphas_cumprod)
    s.set_timesteps(50)
    assert int(s.timesteps[0]) == 981 and int(s.timesteps[-1]) == 1
    # a file WITHOUT clip_sample: diffusers' DDPM class default applies (clip to [-1, 1]) -- Appendix B.1's open item is
    # decided by the checkpoint, not by the engine
    cfg = {k: v for k, v in SD14_SCHEDULER_JSON.items() if k != "clip_sample"}
    s2 = DDPMScheduler.from_pretrained(_write(tmp_path, cfg))
    assert s2.config.clip_sample is True and s2.config.clip_sample_range == 1.0
    d = DDIMScheduler.from_pretrained(_write(tmp_path, SD14_SCHEDULER_JSON))
    assert float(d.final_alpha_cumprod) == float(d.alphas_cumprod[0])   # set_alpha_to_one false, from the file
    with pytest.raises(NotImplementedError):
        DDPMScheduler.from_config(dict(SD14_SCHEDULER_JSON, variance_type="learned_range"))
    with pytest.raises(FileNotFoundError):
        DDPMScheduler.from_pretrained(str(tmp_path / "nowhere"))


def test_weight_discovery_and_config_mapping(tmp_path):
    from safetensors.torch import save_file
    d = tmp_path / "unet"
    d.mkdir()
    sd = {"conv_in.weight": torch.randn(8, 4, 3, 3), "conv_in.bias": torch.randn(8)}
    save_file(sd, str(d / "diffusion_pytorch_model.safetensors"))
    torch.save({k: v * 2 for k, v in sd.items()}, str(d / "diffusion_pytorch_model.fp16.bin"))
    got = ck.load_weights(str(d))
    assert torch.equal(got["conv_in.weight"], sd["conv_in.weight"])
    assert ck.find_weights(str(d), "fp16").endswith(".fp16.bin")
    assert torch.equal(ck.load_weights(str(d), "fp16")["conv_in.bias"], sd["conv_in.bias"] * 2)
    with pytest.raises(FileNotFoundError):
        ck.find_weights(str(tmp_path))
    cfg = {"in_channels": 4, "out_channels": 4, "sample_size": 64, "block_out_channels": [320, 640, 1280, 1280],
           "down_block_types": ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"],
           "up_block_types": ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3, "layers_per_block": 2, "attention_head_dim": 8,
           "cross_attention_dim": 768, "norm_num_groups": 32, "act_fn": "silu", "use_linear_projection": False,
           "_class_name": "UNet2DConditionModel", "norm_eps": 1e-05}
    kw = ck.unet_kwargs(cfg)
    assert kw["block_out_channels"] == (320, 640, 1280, 1280) and "act_fn" not in kw and kw["attention_head_dim"] == 8
    with pytest.raises(NotImplementedError):
        ck.unet_kwargs(dict(cfg, use_linear_projection=True))             # SD-2.x style checkpoints: not this plan
    with pytest.raises(NotImplementedError):
        ck.unet_kwargs(dict(cfg, up_block_types=["CrossAttnUpBlock2D"] * 4))
    assert ck.vae_kwargs({"latent_channels": 4, "block_out_channels": [128, 256, 512, 512], "act_fn": "silu",
                          "scaling_factor": 0.18215, "shift_factor": None})["scaling_factor"] == 0.18215
    assert ck.clip_kwargs({"hidden_size": 768, "hidden_act": "quick_gelu", "projection_dim": 768}) == {"hidden_size": 768}
    assert ck.load_tokenizer(str(tmp_path)) is None
