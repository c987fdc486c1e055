"""Local checkpoint directories: the on-disk format on the input side of the engine.

The reference enters through `DDPMScheduler.from_pretrained(model_id, subfolder="scheduler")` and
`pipeline_func.from_pretrained(model_id, scheduler=scheduler, torch_dtype=weight_dtype, revision="fp16")`
(run_nudity.py:104-122).  There is no network here, so `model_id` must be a LOCAL directory in the diffusers layout:
    unet/config.json             unet/diffusion_pytorch_model.safetensors | .fp16.safetensors | .bin
    vae/config.json              vae/diffusion_pytorch_model.*
    text_encoder/config.json     text_encoder/model.safetensors | pytorch_model.bin
    tokenizer/{vocab.json,merges.txt,...}      (handed to transformers.CLIPTokenizer when the files are there)
    scheduler/scheduler_config.json
Host logic only (files -> dicts); the engine classes upload the tensors.  Configuration values the engine's plans do not
implement are rejected loudly instead of being ignored.
"""
from __future__ import annotations

import json
import os
from typing import Optional

_WEIGHT_NAMES = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "model.safetensors",
                 "model.fp16.safetensors", "diffusion_pytorch_model.bin", "diffusion_pytorch_model.fp16.bin", "pytorch_model.bin",
                 "pytorch_model.fp16.bin")


def read_config(component_dir: str, name: str = "config.json") -> dict:
    path = os.path.join(component_dir, name)
    if not os.path.isfile(path):
        raise FileNotFoundError(f"{path}: not found (expected a diffusers-layout checkpoint directory)")
    with open(path) as f:
        return json.load(f)


def find_weights(component_dir: str, variant: Optional[str] = None) -> str:
    names = _WEIGHT_NAMES
    if variant:                                                       # e.g. "fp16": prefer the matching files
        names = tuple(n for n in names if f".{variant}." in n) + tuple(n for n in names if f".{variant}." not in n)
    for n in names:
        p = os.path.join(component_dir, n)
        if os.path.isfile(p):
            return p
    raise FileNotFoundError(f"no weight file in {component_dir} (looked for {', '.join(names)})")


def load_weights(component_dir: str, variant: Optional[str] = None) -> dict:
    """diffusers-keyed state_dict of one component, as CPU tensors."""
    path = find_weights(component_dir, variant)
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return sd.get("state_dict", sd) if isinstance(sd, dict) else sd


def _require(cfg: dict, what: str, **expected):
    for k, v in expected.items():
        if k in cfg and cfg[k] != v and not (isinstance(v, tuple) and cfg[k] in v):
            raise NotImplementedError(f"{what}: config value {k} = {cfg[k]!r} is not implemented by the engine's plan (needs {v!r})")


def unet_kwargs(cfg: dict) -> dict:
    """UNet2DConditionModel(**kwargs) from unet/config.json (the SD-v1.x family)."""
    _require(cfg, "unet", act_fn="silu", use_linear_projection=False, center_input_sample=False, flip_sin_to_cos=True,
             freq_shift=0, downsample_padding=1, mid_block_scale_factor=1, norm_eps=1e-5, dual_cross_attention=False,
             only_cross_attention=False, class_embed_type=None, upcast_attention=False, resnet_time_scale_shift="default",
             mid_block_type="UNetMidBlock2DCrossAttn", time_embedding_type="positional", addition_embed_type=None)
    down = tuple(cfg.get("down_block_types", ()))
    if any(t not in ("CrossAttnDownBlock2D", "DownBlock2D") for t in down):
        raise NotImplementedError(f"unet: down_block_types {down} are not implemented")
    up = tuple(cfg.get("up_block_types", ()))
    if up and tuple("CrossAttnUpBlock2D" if "CrossAttn" in t else "UpBlock2D" for t in reversed(down)) != up:
        raise NotImplementedError(f"unet: up_block_types {up} do not mirror down_block_types {down}")
    ahd = cfg.get("attention_head_dim", 8)
    if not isinstance(ahd, int):
        raise NotImplementedError("unet: per-level attention_head_dim is not implemented")
    keys = ("in_channels", "out_channels", "sample_size", "block_out_channels", "down_block_types", "layers_per_block",
            "attention_head_dim", "cross_attention_dim", "norm_num_groups")
    out = {k: (tuple(cfg[k]) if isinstance(cfg[k], list) else cfg[k]) for k in keys if k in cfg}
    return out


def vae_kwargs(cfg: dict) -> dict:
    _require(cfg, "vae", act_fn="silu")
    keys = ("in_channels", "out_channels", "latent_channels", "block_out_channels", "layers_per_block", "norm_num_groups",
            "sample_size", "scaling_factor", "shift_factor", "use_quant_conv", "use_post_quant_conv")
    out = {k: (tuple(cfg[k]) if isinstance(cfg[k], list) else cfg[k]) for k in keys if k in cfg and cfg[k] is not None}
    return out


def clip_kwargs(cfg: dict) -> dict:
    _require(cfg, "text_encoder", hidden_act="quick_gelu")
    keys = ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "max_position_embeddings")
    return {k: cfg[k] for k in keys if k in cfg}


def load_tokenizer(model_dir: str):
    """transformers.CLIPTokenizer from `tokenizer/` when its vocabulary files exist; None otherwise (pass tokenizer=...)."""
    d = os.path.join(model_dir, "tokenizer")
    if not (os.path.isfile(os.path.join(d, "vocab.json")) and os.path.isfile(os.path.join(d, "merges.txt"))):
        return None
    from transformers import CLIPTokenizer
    return CLIPTokenizer.from_pretrained(d)
