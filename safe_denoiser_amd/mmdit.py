"""SD3Transformer2DModel (MMDiT) front-end: the call surface the reference's SD-v3 pipelines use
(`self.transformer(hidden_states=, timestep=, encoder_hidden_states=, pooled_projections=, return_dict=False)[0]`,
models/sdv3/safe_denoiser_pipeline.py:1120-1127), executed by libsdn's static launch plan (sdn_mmdit_forward).
Weights are addressed by their diffusers state_dict keys and packed once into the engine layout, exactly as for the UNet.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import torch

from . import _lib
from .unet import P_CONV3X3, P_GEGLU_MAT, P_GEGLU_VEC, P_MAT, P_VEC_F32, UNet2DConditionModel

P_POS_CROP = 5
SD3_MEDIUM = dict(in_channels=16, out_channels=16, sample_size=64, patch_size=2, num_layers=24, num_attention_heads=24,
                  attention_head_dim=64, joint_attention_dim=4096, pooled_projection_dim=2048, pos_embed_max_size=192)


class SD3Transformer2DModel(UNet2DConditionModel):
    """`sample_size` is the LATENT side this plan is built for (64 -> 512x512 images, the reference driver's default,
    run_nudity_sdv3.py:357-358,500; 128 -> 1024x1024)."""

    def __init__(self, text_len: int = 333, dtype=torch.float16, **config):
        if dtype not in (torch.bfloat16, torch.float16):
            raise _lib.SdnError("storage dtype must be torch.bfloat16 or torch.float16")
        self.dtype = dtype
        cfg = dict(SD3_MEDIUM)
        cfg.update(config)
        self.config = SimpleNamespace(**cfg)
        self.in_channels = cfg["in_channels"]
        self.text_len = text_len
        c = _lib.MmditConfig(in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
                             sample_size=cfg["sample_size"], patch_size=cfg["patch_size"], num_layers=cfg["num_layers"],
                             num_heads=cfg["num_attention_heads"], head_dim=cfg["attention_head_dim"],
                             joint_dim=cfg["joint_attention_dim"], pooled_dim=cfg["pooled_projection_dim"],
                             text_len=text_len, time_dim=256, dtype=0 if dtype == torch.bfloat16 else 1)
        h = C.c_void_p()
        _lib.check(_lib.lib().sdn_mmdit_create(C.byref(c), C.byref(h)), "sdn_mmdit_create")
        self._h = h
        self._weights = None
        self._ws = {}
        self._read_manifest()

    # ---- parameters -------------------------------------------------------------------------------------
    def state_dict_shapes(self) -> dict:
        cfg = self.config
        out = {}
        for p in self.manifest:
            k, r, c = p["kind"], p["rows"], p["cols"]
            if k == P_VEC_F32:
                out[p["name"]] = (r,)
            elif k == P_POS_CROP:
                out[p["name"]] = (1, cfg.pos_embed_max_size ** 2, c)
            elif p["name"] == "pos_embed.proj.weight":
                out[p["name"]] = (r, cfg.in_channels, cfg.patch_size, cfg.patch_size)
            else:
                out[p["name"]] = (r, c)
        return out

    def synthetic_state_dict(self, seed: int = 1234) -> dict:
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for name, shape in self.state_dict_shapes().items():
            if name == "pos_embed.pos_embed":
                sd[name] = 0.5 * torch.randn(shape, generator=g)
            elif len(shape) == 1:
                sd[name] = 0.2 * (torch.rand(shape, generator=g) - 0.5)
            else:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
                scale = 0.3 if ("norm1" in name or "norm_out" in name) else 1.0      # keep adaLN modulation moderate
                sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * (3.0 / fan_in) ** 0.5 * scale
        return sd

    def crop_pos_embed(self, pe: torch.Tensor) -> torch.Tensor:
        """PatchEmbed.cropped_pos_embed: centre crop of the [max,max] grid to this plan's token grid."""
        cfg = self.config
        m, hp = cfg.pos_embed_max_size, cfg.sample_size // cfg.patch_size
        top = (m - hp) // 2
        grid = pe.reshape(m, m, -1)
        return grid[top:top + hp, top:top + hp].reshape(hp * hp, -1)

    def pack_state_dict(self, sd: dict) -> torch.Tensor:
        buf = torch.zeros(self.weight_bytes, dtype=torch.uint8)
        for p in self.manifest:
            t = sd[p["name"]].detach().float().cpu()
            k = p["kind"]
            if k == P_POS_CROP:
                t = self.crop_pos_embed(t)
            elif k == P_MAT:
                t = t.reshape(p["rows"], p["cols"])
            raw = t.contiguous().view(torch.uint8) if k == P_VEC_F32 else t.to(self.dtype).contiguous().view(torch.uint8)
            buf[p["offset"]:p["offset"] + raw.numel()] = raw.reshape(-1)
        return buf

    # ---- forward ------------------------------------------------------------------------------------------
    def prepare_text(self, encoder_hidden_states: torch.Tensor) -> torch.Tensor:
        e = encoder_hidden_states
        if e.shape[1] != self.text_len or e.shape[2] != self.config.joint_attention_dim:
            raise _lib.SdnError(f"encoder_hidden_states must be [B,{self.text_len},{self.config.joint_attention_dim}]")
        return e.to(self.dtype).contiguous()

    def max_samples(self) -> int:
        """Largest batch ONE launch plan addresses (31-bit LDS-DMA offsets per operand): the widest 16-bit operand is the feed-forward
        hidden state, tokens x 4 x width x 2 B per sample -- 170 samples at 512^2, 42 at 1024^2 for SD3-medium.  `forward_into` runs
        larger batches as consecutive row blocks (samples do not interact; one handle: this plan keeps no per-text cache)."""
        c = self.config
        tokens = (c.sample_size // c.patch_size) ** 2
        return ((1 << 31) - 1) // (tokens * 4 * c.num_attention_heads * c.attention_head_dim * 2)

    def forward_into(self, sample, timestep, text16, pooled16, out):
        b = sample.shape[0]
        cap = self.max_samples()
        if b > cap:
            if text16.shape[0] != b or pooled16.shape[0] != b or out.shape[0] != b:
                raise _lib.SdnError("batch mismatch between sample, text, pooled projections and out")
            per = max(cap // 2 * 2, 1)                          # (even: a classifier-free-guidance pair's halves stay whole blocks apart)
            for lo in range(0, b, per):
                hi = min(b, lo + per)
                self.forward_into(sample[lo:hi], timestep, text16[lo:hi], pooled16[lo:hi], out[lo:hi])
            return out
        ws = self._workspace(b, sample.device)
        _lib.check(_lib.lib().sdn_mmdit_forward(self._h, _lib.dptr(self._weights), _lib.dptr(sample, torch.float32),
                                                float(timestep), _lib.dptr(text16, self.dtype),
                                                _lib.dptr(pooled16, self.dtype), _lib.dptr(out, torch.float32), b,
                                                _lib.dptr(ws), ws.numel(), _lib.stream_ptr()), "sdn_mmdit_forward")
        return out

    def __call__(self, hidden_states, timestep=None, encoder_hidden_states=None, pooled_projections=None,
                 joint_attention_kwargs=None, return_dict=False, **unused):
        _lib.require_gpu()
        if self._weights is None:
            raise _lib.SdnError("no weights loaded: call load_state_dict() first")
        in_dtype = hidden_states.dtype
        x = hidden_states.float().contiguous()
        s = self.config.sample_size
        if tuple(x.shape[1:]) != (self.config.in_channels, s, s):
            raise _lib.SdnError(f"hidden_states must be [B,{self.config.in_channels},{s},{s}], got {tuple(x.shape)}")
        t = timestep
        if torch.is_tensor(t):
            t = float(t.reshape(-1)[0])                         # the reference broadcasts one t to the batch (:1114)
        e = self.prepare_text(encoder_hidden_states)
        pl = pooled_projections.to(self.dtype).contiguous()
        out = torch.empty((x.shape[0], self.config.out_channels, s, s), dtype=torch.float32, device=x.device)
        self.forward_into(x, t, e, pl, out)
        out = out.to(in_dtype)
        return (out,) if not return_dict else SimpleNamespace(sample=out)
