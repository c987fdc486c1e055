"""AutoencoderKL front-end: `vae.decode(z).sample`, `vae.config.scaling_factor` and the pipelines'
`decode_latents` / `numpy_to_pil` tail (StableDiffusionPipeline.decode_latents as called at
models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:589-596), executed by libsdn's static launch
plan (sdn_vae_decoder_create / sdn_vae_decode).  SURVEY section 8f row 2.

The encoder half serves the proj_ref builder's embed_fn, `vae.encode(x).latent_dist.sample() * scaling_factor`
(run_nudity.py:308 -> RepellencyMethod.project, repellency_methods_threshold.py:54-72): sdn_vae_encoder_create /
sdn_vae_encode / sdn_gaussian_sample.

Weights: the `post_quant_conv.*` + `decoder.*` (and, when present, `quant_conv.*` + `encoder.*`) entries of a diffusers
AutoencoderKL state_dict (SD-v1.4 `vae/`), packed once into the engine layout exactly like the UNet's.  The deprecated attention names of the on-disk checkpoint
(`query/key/value/proj_attn`) are accepted as aliases of `to_q/to_k/to_v/to_out.0`.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import torch

from . import _lib
from .unet import UNet2DConditionModel

SD14_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                       layers_per_block=2, norm_num_groups=32, sample_size=512, scaling_factor=0.18215, shift_factor=0.0,
                       use_quant_conv=True, use_post_quant_conv=True)
# SD-v3's 16-channel VAE (models/sdv3/safe_denoiser_pipeline.py:1196: latents / scaling_factor + shift_factor; no quant convs)
SD3_VAE_CONFIG = dict(SD14_VAE_CONFIG, latent_channels=16, scaling_factor=1.5305, shift_factor=0.0609, use_quant_conv=False,
                      use_post_quant_conv=False)

_DEPRECATED_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


class DecoderOutput:
    __slots__ = ("sample",)

    def __init__(self, sample):
        self.sample = sample


class DiagonalGaussianDistribution:
    """`latent_dist` of AutoencoderKLOutput: moments [B, 2L, S, S] = (mean | logvar), logvar clamped to [-30, 20]."""

    def __init__(self, moments: torch.Tensor):
        self.parameters = moments
        self.mean, lv = moments.chunk(2, dim=1)
        self.logvar = lv.clamp(-30.0, 20.0)

    @property
    def std(self):
        return torch.exp(0.5 * self.logvar)

    def _draw(self, noise, scale):
        m = self.parameters
        b, c2, h, w = m.shape
        out = torch.empty((b, c2 // 2, h, w), dtype=torch.float32, device=m.device)
        _lib.check(_lib.lib().sdn_gaussian_sample(_lib.dptr(m, torch.float32), None if noise is None else _lib.dptr(noise, torch.float32),
                                                  b, c2 // 2, h * w, float(scale), _lib.dptr(out, torch.float32), _lib.stream_ptr()),
                   "sdn_gaussian_sample")
        return out

    def sample(self, generator=None, scale: float = 1.0) -> torch.Tensor:
        """mean + std * randn (diffusers draws with randn_tensor(mean.shape, generator) on the parameters' device)."""
        m = self.parameters
        noise = torch.randn((m.shape[0], m.shape[1] // 2) + tuple(m.shape[2:]), generator=generator, device=m.device,
                            dtype=torch.float32)
        return self._draw(noise, scale)

    def mode(self, scale: float = 1.0) -> torch.Tensor:
        return self._draw(None, scale)


class EncoderOutput:
    __slots__ = ("latent_dist",)

    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class AutoencoderKL(UNet2DConditionModel):
    """AutoencoderKL.  `sample_size` is the IMAGE side (diffusers' meaning); the latent side is
    sample_size / 2**(levels-1).  The object itself is the decoder half; `.encoder_half` (built on first use or when
    the state_dict carries `encoder.*`) is the same class in the encoder role."""

    MAX_CHUNK = 8                      # images per plan invocation; sdn_vae_* chunk larger batches themselves (32-bit DMA offsets)

    def __init__(self, dtype=torch.bfloat16, _role: str = "decoder", **config):
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
        # images per plan invocation: the C side bounds batch * side^2 * max(channels) * 2 bytes by 2^32 (DMA offsets)
        self.MAX_CHUNK = max(1, min(8, (2 ** 32 - 1) // (cfg["sample_size"] ** 2 * max(boc) * 2)))
        c = _lib.VaeConfig(latent_channels=cfg["latent_channels"], out_channels=cfg["out_channels"],
                           sample_size=self.latent_size, n_levels=n,
                           block_out_channels=(C.c_int32 * 4)(*(boc + [0] * (4 - n))),
                           layers_per_block=cfg["layers_per_block"], norm_groups=cfg["norm_num_groups"],
                           dtype=0 if dtype == torch.bfloat16 else 1)
        h = C.c_void_p()
        self._role = _role
        self._user_config = dict(config)
        self.encoder_half = None
        if _role == "encoder":
            _lib.check(_lib.lib().sdn_vae_encoder_create(C.byref(c), C.byref(h)), "sdn_vae_encoder_create")
        else:
            _lib.check(_lib.lib().sdn_vae_decoder_create(C.byref(c), C.byref(h)), "sdn_vae_decoder_create")
        self._h = h
        self._weights = None
        self._ws = {}
        self._read_manifest()

    # ---- parameters ---------------------------------------------------------------------------------
    @staticmethod
    def _is_norm_param(name: str) -> bool:
        return "norm" in name.split(".")[-2]

    def state_dict_shapes(self) -> dict:
        out = super().state_dict_shapes()
        L = self.config.latent_channels
        if self._role == "encoder":
            out["quant_conv.weight"] = (2 * L, 2 * L, 1, 1)
        else:
            out["post_quant_conv.weight"] = (L, L, 1, 1)
        return out

    def _encoder(self):
        if self.encoder_half is None:
            self.encoder_half = AutoencoderKL(dtype=self.dtype, _role="encoder", **self._user_config)
        return self.encoder_half

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
        # VAEs without (post_)quant_conv (SD-v3): the plan's 1x1 mixing stage gets the identity
        L = self.config.latent_channels
        if self._role == "decoder" and not self.config.use_post_quant_conv:
            sd["post_quant_conv.weight"], sd["post_quant_conv.bias"] = torch.eye(L).reshape(L, L, 1, 1), torch.zeros(L)
        if self._role == "encoder" and not self.config.use_quant_conv:
            sd["quant_conv.weight"], sd["quant_conv.bias"] = torch.eye(2 * L).reshape(2 * L, 2 * L, 1, 1), torch.zeros(2 * L)
        qk = "quant_conv.weight" if self._role == "encoder" else "post_quant_conv.weight"
        sd[qk] = sd[qk].reshape(-1)                                                   # [C, C, 1, 1] -> fp32 vector
        for k in list(sd):                                                            # deprecated linears stored as 1x1 convs
            if "attentions" in k and k.endswith("weight") and sd[k].dim() == 4:
                sd[k] = sd[k].reshape(sd[k].shape[0], sd[k].shape[1])
        return super().pack_state_dict(sd)

    def load_state_dict(self, sd: dict, device="cuda"):
        sd = dict(self._canonical(sd))
        L = self.config.latent_channels
        if self._role == "decoder" and not self.config.use_post_quant_conv:
            sd.setdefault("post_quant_conv.weight", torch.eye(L).reshape(L, L, 1, 1)); sd.setdefault("post_quant_conv.bias", torch.zeros(L))
        if self._role == "encoder" and not self.config.use_quant_conv:
            sd.setdefault("quant_conv.weight", torch.eye(2 * L).reshape(2 * L, 2 * L, 1, 1)); sd.setdefault("quant_conv.bias", torch.zeros(2 * L))
        if self._role == "decoder" and "encoder.conv_in.weight" in sd:
            self._encoder().load_state_dict(sd, device)
        return super().load_state_dict(sd, device)

    def synthetic_state_dict(self, seed: int = 1234, with_encoder: bool = False) -> dict:
        sd = super().synthetic_state_dict(seed)
        if with_encoder and self._role == "decoder":
            sd.update(self._encoder().synthetic_state_dict(seed + 1))
        return sd

    # ---- encode ---------------------------------------------------------------------------------------
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """quant_conv(encoder(x)) -> latent_dist; x = images [B, 3, H, W] in [-1, 1]."""
        enc = self._encoder() if self._role == "decoder" else self
        _lib.require_gpu()
        if enc._weights is None:
            raise _lib.SdnError("no encoder weights loaded: the state_dict given to load_state_dict() had no `encoder.*` keys")
        x = x.float().contiguous()
        side = enc.latent_size * enc.up_factor
        if tuple(x.shape[1:]) != (enc.config.out_channels, side, side):
            raise _lib.SdnError(f"images must be [B,{enc.config.out_channels},{side},{side}], got {tuple(x.shape)}")
        L, s = enc.config.latent_channels, enc.latent_size
        mom = torch.empty((x.shape[0], 2 * L, s, s), dtype=torch.float32, device=x.device)
        ws = enc._workspace(x.shape[0], x.device)                      # the entry point chunks large batches itself
        _lib.check(_lib.lib().sdn_vae_encode(enc._h, _lib.dptr(enc._weights), _lib.dptr(x, torch.float32),
                                             _lib.dptr(mom, torch.float32), x.shape[0], _lib.dptr(ws), ws.numel(),
                                             _lib.stream_ptr()), "sdn_vae_encode")
        dist = DiagonalGaussianDistribution(mom)
        return EncoderOutput(dist) if return_dict else (dist,)

    def embed_fn(self, generator=None):
        """The reference's `embed_fn` (run_nudity.py:308): x -> vae.encode(x).latent_dist.sample() * scaling_factor."""
        return lambda x: self.encode(x).latent_dist.sample(generator, scale=self.config.scaling_factor)

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
        self._decode_into(z, latent_scale, out)                         # any batch: sdn_vae_decode chunks internally
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

    def _unscale(self, latents: torch.Tensor) -> torch.Tensor:
        """latents / scaling_factor + shift_factor (SD-v3, safe_denoiser_pipeline.py:1196); the division alone is folded
        into the plan's first kernel, a non-zero shift is applied here."""
        sh = getattr(self.config, "shift_factor", 0.0) or 0.0
        return latents if sh == 0.0 else latents.float() + sh * self.config.scaling_factor      # (z + sh*s)/s = z/s + sh

    def decode_latents(self, latents: torch.Tensor):
        """The reference pipelines' decode_latents: NHWC float32 numpy in [0, 1]."""
        image = self.decode(self._unscale(latents), latent_scale=1.0 / self.config.scaling_factor).sample
        return self.postprocess(image).cpu().numpy()

    def decode_latents_uint8(self, latents: torch.Tensor) -> torch.Tensor:
        """decode_latents + numpy_to_pil's uint8 conversion, left on the device: [B, H, W, 3] uint8."""
        image = self.decode(self._unscale(latents), latent_scale=1.0 / self.config.scaling_factor).sample
        return self.postprocess(image, uint8=True)

    def __call__(self, *a, **k):
        raise _lib.SdnError("AutoencoderKL front-end exposes encode() / decode() / decode_latents(), not the autoencoding forward")
