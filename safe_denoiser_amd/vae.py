"""AutoencoderKL (decoder half) front-end: `vae.decode(z).sample`, `vae.config.scaling_factor` and the pipelines'
`decode_latents` / `numpy_to_pil` tail (StableDiffusionPipeline.decode_latents as called at
models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:589-596), executed by libsdn's static launch
plan (sdn_vae_decoder_create / sdn_vae_decode).  SURVEY section 8f row 2.

Weights: the `post_quant_conv.*` and `decoder.*` entries of a diffusers AutoencoderKL state_dict (SD-v1.4 `vae/`), packed
once into the engine layout exactly like the UNet's.  The deprecated attention names of the on-disk checkpoint
(`query/key/value/proj_attn`) are accepted as aliases of `to_q/to_k/to_v/to_out.0`.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import torch

from . import _lib
from .unet import UNet2DConditionModel

SD14_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                       layers_per_block=2, norm_num_groups=32, sample_size=512, scaling_factor=0.18215)

_DEPRECATED_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


class DecoderOutput:
    __slots__ = ("sample",)

    def __init__(self, sample):
        self.sample = sample


class AutoencoderKL(UNet2DConditionModel):
    """Decoder-only AutoencoderKL.  `sample_size` is the IMAGE side (diffusers' meaning); the latent side is
    sample_size / 2**(levels-1)."""

    MAX_CHUNK = 8                      # images per sdn_vae_decode call (32-bit offsets bound it at 15 for 512 x 512)

    def __init__(self, dtype=torch.bfloat16, **config):
        if dtype not in (torch.bfloat16, torch.float16):
            raise _lib.SdnError("storage dtype must be torch.bfloat16 or torch.float16")
        self.dtype = dtype
        cfg = dict(SD14_VAE_CONFIG)
        cfg.update(config)
        self.config = SimpleNamespace(**cfg)
        boc = list(cfg["block_out_channels"])
        n = len(boc)
        self.up_factor = 2 ** (n - 1)
        if cfg["sample_size"] % self.up_factor:
            raise _lib.SdnError("sample_size must be a multiple of 2**(levels-1)")
        self.latent_size = cfg["sample_size"] // self.up_factor
        c = _lib.VaeConfig(latent_channels=cfg["latent_channels"], out_channels=cfg["out_channels"],
                           sample_size=self.latent_size, n_levels=n,
                           block_out_channels=(C.c_int32 * 4)(*(boc + [0] * (4 - n))),
                           layers_per_block=cfg["layers_per_block"], norm_groups=cfg["norm_num_groups"],
                           dtype=0 if dtype == torch.bfloat16 else 1)
        h = C.c_void_p()
        _lib.check(_lib.lib().sdn_vae_decoder_create(C.byref(c), C.byref(h)), "sdn_vae_decoder_create")
        self._h = h
        self._weights = None
        self._ws = {}
        self.manifest = []
        info = _lib.ParamInfo()
        for i in range(_lib.lib().sdn_unet_param_count(h)):
            _lib.check(_lib.lib().sdn_unet_param_info(h, i, C.byref(info)), "sdn_unet_param_info")
            self.manifest.append(dict(name=info.name.decode(), kind=info.kind, rows=info.rows, cols=info.cols,
                                      rows_padded=info.rows_padded, offset=info.offset))
        self.weight_bytes = _lib.lib().sdn_unet_weight_bytes(h)

    # ---- parameters ---------------------------------------------------------------------------------
    @staticmethod
    def _is_norm_param(name: str) -> bool:
        return "norm" in name.split(".")[-2]

    def state_dict_shapes(self) -> dict:
        out = super().state_dict_shapes()
        L = self.config.latent_channels
        out["post_quant_conv.weight"] = (L, L, 1, 1)
        return out

    @staticmethod
    def _canonical(sd: dict) -> dict:
        out = {}
        for k, v in sd.items():
            parts = k.split(".")
            if "attentions" in parts:
                for old, new in _DEPRECATED_ATTN.items():
                    if parts[-2] == old:
                        k = ".".join(parts[:-2] + [new, parts[-1]])
            out[k] = v
        return out

    def pack_state_dict(self, sd: dict) -> torch.Tensor:
        sd = self._canonical(sd)
        sd = dict(sd)
        sd["post_quant_conv.weight"] = sd["post_quant_conv.weight"].reshape(-1)      # [L, L, 1, 1] -> fp32 vector
        for k in list(sd):                                                            # deprecated linears stored as 1x1 convs
            if "attentions" in k and k.endswith("weight") and sd[k].dim() == 4:
                sd[k] = sd[k].reshape(sd[k].shape[0], sd[k].shape[1])
        return super().pack_state_dict(sd)

    def load_state_dict(self, sd: dict, device="cuda"):
        return super().load_state_dict(self._canonical(sd), device)

    # ---- decode ---------------------------------------------------------------------------------------
    def _decode_into(self, z: torch.Tensor, latent_scale: float, out: torch.Tensor):
        b = z.shape[0]
        ws = self._workspace(b, z.device)
        _lib.check(_lib.lib().sdn_vae_decode(self._h, _lib.dptr(self._weights), _lib.dptr(z, torch.float32), float(latent_scale),
                                             _lib.dptr(out, torch.float32), b, _lib.dptr(ws), ws.numel(), _lib.stream_ptr()),
                   "sdn_vae_decode")

    def decode(self, z: torch.Tensor, return_dict: bool = True, latent_scale: float = 1.0, **unused):
        """decoder(post_quant_conv(latent_scale * z)) -> [B, 3, H, W] fp32 (raw, nominally in [-1, 1])."""
        _lib.require_gpu()
        if self._weights is None:
            raise _lib.SdnError("no weights loaded: call load_state_dict() first")
        z = z.float().contiguous()
        s = self.latent_size
        if tuple(z.shape[1:]) != (self.config.latent_channels, s, s):
            raise _lib.SdnError(f"latents must be [B,{self.config.latent_channels},{s},{s}], got {tuple(z.shape)}")
        side = s * self.up_factor
        out = torch.empty((z.shape[0], self.config.out_channels, side, side), dtype=torch.float32, device=z.device)
        for lo in range(0, z.shape[0], self.MAX_CHUNK):
            hi = min(lo + self.MAX_CHUNK, z.shape[0])
            self._decode_into(z[lo:hi], latent_scale, out[lo:hi])
        return DecoderOutput(out) if return_dict else (out,)

    def postprocess(self, image: torch.Tensor, uint8: bool = False) -> torch.Tensor:
        """(image / 2 + 0.5).clamp(0, 1) -> NHWC fp32, or round(255 x) uint8 NHWC (numpy_to_pil's conversion)."""
        _lib.require_gpu()
        image = image.float().contiguous()
        b, c, h, w = image.shape
        out = torch.empty((b, h, w, c), dtype=torch.uint8 if uint8 else torch.float32, device=image.device)
        _lib.check(_lib.lib().sdn_image_postprocess(_lib.dptr(image, torch.float32), b, c, h, w,
                                                    None if uint8 else _lib.dptr(out, torch.float32),
                                                    _lib.dptr(out, torch.uint8) if uint8 else None, _lib.stream_ptr()),
                   "sdn_image_postprocess")
        return out

    def decode_latents(self, latents: torch.Tensor):
        """The reference pipelines' decode_latents: NHWC float32 numpy in [0, 1]."""
        image = self.decode(latents, latent_scale=1.0 / self.config.scaling_factor).sample
        return self.postprocess(image).cpu().numpy()

    def decode_latents_uint8(self, latents: torch.Tensor) -> torch.Tensor:
        """decode_latents + numpy_to_pil's uint8 conversion, left on the device: [B, H, W, 3] uint8."""
        image = self.decode(latents, latent_scale=1.0 / self.config.scaling_factor).sample
        return self.postprocess(image, uint8=True)

    def __call__(self, *a, **k):
        raise _lib.SdnError("AutoencoderKL front-end exposes decode() / decode_latents(); the encoder is not built yet")
