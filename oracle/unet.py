"""CPU oracle: SD-v1.4 UNet2DConditionModel forward, plain torch ops, NCHW.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

PARITY UNPINNED at the reference level: the live network is diffusers==0.29.0's UNet2DConditionModel (absent here;
requirements.txt:3) and the reference holds no tests for it.  Wiring follows the reference's vendored (dead) copies,
  models/unet.py:683-932 (forward: time embed :764-788, conv_in :840, down :842-857, mid :871-879, up :886-919,
  tail :922-927), models/unet_2d_blocks.py:1266-1335 / :1389-1426 / :872-924 / :2507-2589 / :2642-2704,
  models/transformer_2d.py:239-359 (BasicTransformerBlock), :505-540,810-858 (Transformer2DModel continuous path),
and the published diffusers-0.29.0 leaf definitions (ResnetBlock2D, Attention/AttnProcessor2_0, GEGLU FeedForward,
Timesteps, TimestepEmbedding, Down/Upsample2D) restated in SURVEY.md appendix A.  Parameters are addressed by their
diffusers state_dict keys, so a real checkpoint would drop in.  Two structural checks pin the wiring against public
facts about the architecture: 686 state_dict keys / 859,520,964 parameters (tests/test_unet_host.py).

``act_dtype`` = torch.bfloat16 emulates the engine's storage precision: every tensor the HIP engine writes to HBM as
bf16 is rounded to bf16 at the same point (weights too); all arithmetic in between stays fp32.  ``None`` = pure fp32.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

SD14 = dict(in_channels=4, out_channels=4, sample_size=64, block_out_channels=(320, 640, 1280, 1280),
            level_has_attn=(True, True, True, False), layers_per_block=2, n_heads=8, cross_dim=768, norm_groups=32)


class OracleUNet:
    # classes of rounding points (the precision ablation, tools/precision_ablation.py, promotes one class at a time):
    #   "w" matrices; "text"/"temb" the conditioning inputs; "norm" GroupNorm/LayerNorm outputs (= GEMM A operands);
    #   "stream" the residual / skip stream (resnet, attention, feed-forward and transformer outputs, conv_in, re-sampling
    #   convs); "inner" tensors between two GEMMs inside a branch (conv1, proj_in, the 1x1 shortcut); "qkv" q, k, v and the
    #   attention output; "ff" the GEGLU hidden activation; "opnd" the stream where a GEMM consumes it WITHOUT a norm in
    #   between (1x1 shortcut, re-sampling convs; idempotent after "stream" when both round to the same type)
    KINDS = ("w", "text", "temb", "norm", "stream", "inner", "qkv", "ff", "opnd")

    def __init__(self, state_dict: dict, config: dict | None = None, act_dtype=None, q_map: dict | None = None,
                 device=None):
        """q_map: {kind: rounding} overrides `act_dtype` per class; a rounding is None (keep fp32), a torch dtype, or a
        callable x -> x (e.g. a two-term bf16 split)."""
        self.cfg = dict(SD14)
        if config:
            self.cfg.update(config)
        self.q_dtype = act_dtype
        self.device = device
        self.q_map = {k: act_dtype for k in self.KINDS}
        if q_map:
            self.q_map.update(q_map)
        self.sd = {}
        for k, v in state_dict.items():
            v = v.detach().float()
            if device is not None:
                v = v.to(device)
            if v.dim() > 1:                                      # matrices are stored bf16 by the engine; vectors f32
                v = self._round(v, self.q_map["w"])
            self.sd[k] = v

    # ---- helpers ---------------------------------------------------------------------------------
    @staticmethod
    def _round(x, how):
        if how is None:
            return x
        if callable(how):
            return how(x)
        return x.to(how).float()

    def q(self, x, kind="stream"):
        return self._round(x, self.q_map[kind])

    def P(self, name):
        return self.sd[name]

    def timestep_features(self, t: float, batch: int, dim: int) -> torch.Tensor:
        """Timesteps(dim, flip_sin_to_cos=True, freq_shift=0): [cos | sin] (models/unet.py:764-786)."""
        half = dim // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        ang = torch.full((batch, 1), float(t), dtype=torch.float32) * freqs[None]
        if self.device is not None:
            freqs, ang = freqs.to(self.device), ang.to(self.device)
        return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)

    def resnet(self, pfx, x, semb):
        """ResnetBlock2D; `semb` = SiLU(temb) (temb enters only through SiLU -> Linear)."""
        g = self.cfg["norm_groups"]
        h = self.q(F.silu(F.group_norm(x, g, self.P(pfx + ".norm1.weight"), self.P(pfx + ".norm1.bias"), eps=1e-5)), "norm")
        tp = F.linear(semb, self.P(pfx + ".time_emb_proj.weight"), self.P(pfx + ".time_emb_proj.bias"))   # stays f32
        h = self.q(F.conv2d(h, self.P(pfx + ".conv1.weight"), self.P(pfx + ".conv1.bias"), padding=1) + tp[:, :, None, None], "inner")
        h = self.q(F.silu(F.group_norm(h, g, self.P(pfx + ".norm2.weight"), self.P(pfx + ".norm2.bias"), eps=1e-5)), "norm")
        if (pfx + ".conv_shortcut.weight") in self.sd:
            sc = self.q(F.conv2d(self.q(x, "opnd"), self.P(pfx + ".conv_shortcut.weight"), self.P(pfx + ".conv_shortcut.bias")), "inner")
        else:
            sc = x
        return self.q(F.conv2d(h, self.P(pfx + ".conv2.weight"), self.P(pfx + ".conv2.bias"), padding=1) + sc)

    def attention(self, pfx, x, ctx):
        """Attention + AttnProcessor2_0: bias-free q/k/v, 8 heads, SDPA scale d^-1/2, to_out with bias.
        Returns the to_out output WITHOUT the residual."""
        nh = self.cfg["n_heads"]
        b, n, c = x.shape
        q = self.q(F.linear(x, self.P(pfx + ".to_q.weight")), "qkv")
        k = self.q(F.linear(ctx, self.P(pfx + ".to_k.weight")), "qkv")
        v = self.q(F.linear(ctx, self.P(pfx + ".to_v.weight")), "qkv")
        d = c // nh
        q, k, v = (t.reshape(b, -1, nh, d).transpose(1, 2) for t in (q, k, v))
        a = F.scaled_dot_product_attention(q, k, v)
        a = self.q(a.transpose(1, 2).reshape(b, n, c), "qkv")
        return F.linear(a, self.P(pfx + ".to_out.0.weight"), self.P(pfx + ".to_out.0.bias"))

    def transformer(self, pfx, x, text):
        """Transformer2DModel (continuous, use_linear_projection=False) with one BasicTransformerBlock."""
        g = self.cfg["norm_groups"]
        b, c, hh, ww = x.shape
        tb = pfx + ".transformer_blocks.0"
        h = self.q(F.group_norm(x, g, self.P(pfx + ".norm.weight"), self.P(pfx + ".norm.bias"), eps=1e-6), "norm")
        h = self.q(F.conv2d(h, self.P(pfx + ".proj_in.weight"), self.P(pfx + ".proj_in.bias")), "stream")
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        ln = self.q(F.layer_norm(h, (c,), self.P(tb + ".norm1.weight"), self.P(tb + ".norm1.bias"), eps=1e-5), "norm")
        h = self.q(self.attention(tb + ".attn1", ln, ln) + h)
        ln = self.q(F.layer_norm(h, (c,), self.P(tb + ".norm2.weight"), self.P(tb + ".norm2.bias"), eps=1e-5), "norm")
        h = self.q(self.attention(tb + ".attn2", ln, text) + h)
        ln = self.q(F.layer_norm(h, (c,), self.P(tb + ".norm3.weight"), self.P(tb + ".norm3.bias"), eps=1e-5), "norm")
        proj = F.linear(ln, self.P(tb + ".ff.net.0.proj.weight"), self.P(tb + ".ff.net.0.proj.bias"))
        val, gate = proj.chunk(2, dim=-1)
        ff = self.q(val * F.gelu(gate), "ff")                                   # exact (erf) GELU
        h = self.q(F.linear(ff, self.P(tb + ".ff.net.2.weight"), self.P(tb + ".ff.net.2.bias")) + h)
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
        return self.q(F.conv2d(h, self.P(pfx + ".proj_out.weight"), self.P(pfx + ".proj_out.bias")) + x)

    # ---- forward -------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, sample: torch.Tensor, timestep: float, encoder_hidden_states: torch.Tensor) -> torch.Tensor:
        c = self.cfg
        boc = c["block_out_channels"]
        nl = len(boc)
        b = sample.shape[0]
        text = self.q(encoder_hidden_states.float(), "text")
        te = self.q(self.timestep_features(timestep, b, boc[0]), "temb")
        te = self.q(F.silu(F.linear(te, self.P("time_embedding.linear_1.weight"), self.P("time_embedding.linear_1.bias"))), "temb")
        semb = self.q(F.silu(F.linear(te, self.P("time_embedding.linear_2.weight"), self.P("time_embedding.linear_2.bias"))), "temb")

        h = self.q(F.conv2d(sample.float(), self.P("conv_in.weight"), self.P("conv_in.bias"), padding=1))
        skips = [h]
        for i in range(nl):
            for j in range(c["layers_per_block"]):
                h = self.resnet(f"down_blocks.{i}.resnets.{j}", h, semb)
                if c["level_has_attn"][i]:
                    h = self.transformer(f"down_blocks.{i}.attentions.{j}", h, text)
                skips.append(h)
            if i + 1 < nl:
                p = f"down_blocks.{i}.downsamplers.0.conv"
                h = self.q(F.conv2d(self.q(h, "opnd"), self.P(p + ".weight"), self.P(p + ".bias"), stride=2, padding=1))
                skips.append(h)
        h = self.resnet("mid_block.resnets.0", h, semb)
        h = self.transformer("mid_block.attentions.0", h, text)
        h = self.resnet("mid_block.resnets.1", h, semb)
        for i in range(nl):
            lvl = nl - 1 - i
            for j in range(c["layers_per_block"] + 1):
                h = torch.cat([h, skips.pop()], dim=1)                   # models/unet_2d_blocks.py:2546,2679
                h = self.resnet(f"up_blocks.{i}.resnets.{j}", h, semb)
                if c["level_has_attn"][lvl]:
                    h = self.transformer(f"up_blocks.{i}.attentions.{j}", h, text)
            if i + 1 < nl:
                p = f"up_blocks.{i}.upsamplers.0.conv"
                h = F.interpolate(self.q(h, "opnd"), scale_factor=2.0, mode="nearest")
                h = self.q(F.conv2d(h, self.P(p + ".weight"), self.P(p + ".bias"), padding=1))
        h = self.q(F.silu(F.group_norm(h, c["norm_groups"], self.P("conv_norm_out.weight"), self.P("conv_norm_out.bias"), eps=1e-5)), "norm")
        return F.conv2d(h, self.P("conv_out.weight"), self.P("conv_out.bias"), padding=1)
