"""CLIPTextModel front-end: `text_encoder(input_ids, attention_mask=None)[0]` as the reference's pipelines call it
(models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:197,225,287,333), executed by libsdn's launch
plan (sdn_clip_create / sdn_clip_forward).  SURVEY section 8f row 4.

Weights: a transformers CLIPTextModel state_dict (keys with or without the `text_model.` prefix), packed once into the
engine layout.  The tokenizer stays with the caller (its vocabulary files are not part of this engine): pass token ids.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import torch

from . import _lib
from .unet import UNet2DConditionModel

SD14_CLIP_CONFIG = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                        num_attention_heads=12, max_position_embeddings=77)


class TextEncoderOutput(tuple):
    """(last_hidden_state, pooler_output) with attribute access, like transformers' BaseModelOutputWithPooling."""

    def __new__(cls, last_hidden_state, pooler_output):
        o = super().__new__(cls, (last_hidden_state, pooler_output))
        o.last_hidden_state, o.pooler_output = last_hidden_state, pooler_output
        return o


class CLIPTextModel(UNet2DConditionModel):
    def __init__(self, dtype=torch.bfloat16, precision: str | None = None, **config):
        """dtype = bf16 / fp16 storage, or torch.float32 = the plan's fp32 storage mode on the f32-input matrix cores;
        precision = "bf16x3" = fp32 storage with split-operand GEMMs on the bf16 matrix cores (the UNet's tolerance-meeting
        mode).  The reference loads the text encoder in fp32 with the rest of the pipeline (run_nudity.py:277 -> load_sd(...,
        torch.float32)); its hidden states feed every cross-attention AND the SAFREE decisions (trigger-token mask, beta ->
        step count), so the fp32-storage modes are what a seed-for-seed comparison from token ids needs."""
        if precision not in (None, "fp32", "bf16x3"):
            raise _lib.SdnError('precision must be None, "fp32" or "bf16x3"')
        if precision is not None:
            dtype = torch.float32
        if dtype not in (torch.bfloat16, torch.float16, torch.float32):
            raise _lib.SdnError("storage dtype must be torch.bfloat16, torch.float16 or torch.float32")
        self.dtype = dtype
        self.precision = precision or ("fp32" if dtype == torch.float32 else None)
        self.latent_repeat = 1
        cfg = dict(SD14_CLIP_CONFIG)
        cfg.update(config)
        self.config = SimpleNamespace(**cfg)
        c = _lib.ClipConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"],
                            intermediate_size=cfg["intermediate_size"], num_layers=cfg["num_hidden_layers"],
                            num_heads=cfg["num_attention_heads"], max_position_embeddings=cfg["max_position_embeddings"],
                            dtype=3 if self.precision == "bf16x3" else {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}[dtype])
        h = C.c_void_p()
        _lib.check(_lib.lib().sdn_clip_create(C.byref(c), C.byref(h)), "sdn_clip_create")
        self._h = h
        self._weights = None
        self._ws = {}
        self._read_manifest()

    def state_dict_shapes(self) -> dict:
        return {p["name"]: ((p["rows"],) if p["cols"] == 0 else (p["rows"], p["cols"])) for p in self.manifest}

    @staticmethod
    def _is_norm_param(name: str) -> bool:
        return "norm" in name.split(".")[-2]

    @staticmethod
    def _canonical(sd: dict) -> dict:
        return {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}

    def pack_state_dict(self, sd: dict) -> torch.Tensor:
        return super().pack_state_dict(self._canonical(sd))

    def load_state_dict(self, sd: dict, device="cuda"):
        return super().load_state_dict(self._canonical(sd), device)

    def __call__(self, input_ids: torch.Tensor, attention_mask: torch.Tensor | None = None, **unused):
        _lib.require_gpu()
        if self._weights is None:
            raise _lib.SdnError("no weights loaded: call load_state_dict() first")
        n = self.config.max_position_embeddings
        if input_ids.dim() != 2 or input_ids.shape[1] != n:
            raise _lib.SdnError(f"input_ids must be [B,{n}] (tokenizer padding='max_length'), got {tuple(input_ids.shape)}")
        ids = input_ids.to(torch.int32).contiguous()
        mask = None if attention_mask is None else attention_mask.to(device=ids.device, dtype=torch.int32).contiguous()
        if mask is not None and tuple(mask.shape) != tuple(ids.shape):
            raise _lib.SdnError("attention_mask must have the shape of input_ids")
        b = ids.shape[0]
        out = torch.empty((b, n, self.config.hidden_size), dtype=self.dtype, device=ids.device)
        ws = self._workspace(b, ids.device)
        _lib.check(_lib.lib().sdn_clip_forward(self._h, _lib.dptr(self._weights), _lib.dptr(ids, torch.int32),
                                               None if mask is None else _lib.dptr(mask, torch.int32), _lib.dptr(out, self.dtype),
                                               b, _lib.dptr(ws), ws.numel(), _lib.stream_ptr()), "sdn_clip_forward")
        # pooled = features at the EOT token = the highest id of each sequence (CLIPTextTransformer.forward)
        pooled = out[torch.arange(b, device=ids.device), ids.argmax(dim=-1)]
        return TextEncoderOutput(out, pooled)
