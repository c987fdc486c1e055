"""CPU oracle: AutoencoderKL decoder of SD-v1.4, plain torch ops, NCHW.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

PARITY UNPINNED at the reference level: the live module is diffusers==0.29.0's AutoencoderKL (absent here;
requirements.txt:3); the reference only CALLS it -- `self.decode_latents(latents)` at
models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:589 (StableDiffusionPipeline's inherited
method) and the explicit copy in modified_sld_pipeline_threshold_time.py:539-545:
    latents = 1 / 0.18215 * latents; image = vae.decode(latents).sample; image = (image / 2 + 0.5).clamp(0, 1);
    image.cpu().permute(0, 2, 3, 1).float().numpy()           and numpy_to_pil: (images * 255).round().astype(uint8)
The published diffusers-0.29.0 definitions restated here (vae/config.json of CompVis/stable-diffusion-v1-4:
block_out_channels (128, 256, 512, 512), layers_per_block 2, latent_channels 4, norm_num_groups 32, act silu):
  AutoencoderKL.decode(z) = Decoder(post_quant_conv(z)),  post_quant_conv = Conv2d(4, 4, 1)
  Decoder: conv_in Conv2d(4, 512, 3, pad 1) -> UNetMidBlock2D[ResnetBlock2D, Attention, ResnetBlock2D]
           -> 4 x UpDecoderBlock2D (3 resnets each; channels 512, 512, 256, 128; Upsample2D = nearest 2x + conv3x3 on all
              but the last) -> GroupNorm(32, 128, eps 1e-6) -> SiLU -> conv_out Conv2d(128, 3, 3, pad 1)
  ResnetBlock2D(temb=None, eps 1e-6): GN -> SiLU -> conv3x3 -> GN -> SiLU -> conv3x3, + (1x1 conv_shortcut if Cin != Cout)
  Attention(512, heads 1, dim_head 512, bias, residual_connection, GroupNorm(32, eps 1e-6), rescale_output_factor 1):
      h = GN(x) as tokens; softmax(q k^T / sqrt(512)) v; to_out; + x
A structural check pins the wiring against a public fact: the decoder holds 49,490,179 parameters and post_quant_conv
20 (tests/test_oracle_vae.py).

``act_dtype`` emulates the engine's 16-bit storage points (weights of matrices too); None = pure fp32.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SD14_VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                norm_groups=32, scaling_factor=0.18215)


def decoder_state_dict_shapes(cfg: dict | None = None) -> dict:
    """diffusers key -> shape of the decoder half (+ post_quant_conv)."""
    c = dict(SD14_VAE)
    if cfg:
        c.update(cfg)
    boc, L = list(c["block_out_channels"]), c["latent_channels"]
    top = boc[-1]
    out = {"post_quant_conv.weight": (L, L, 1, 1), "post_quant_conv.bias": (L,),
           "decoder.conv_in.weight": (top, L, 3, 3), "decoder.conv_in.bias": (top,)}

    def resnet(pfx, cin, cout):
        out[pfx + ".norm1.weight"] = (cin,); out[pfx + ".norm1.bias"] = (cin,)
        out[pfx + ".conv1.weight"] = (cout, cin, 3, 3); out[pfx + ".conv1.bias"] = (cout,)
        out[pfx + ".norm2.weight"] = (cout,); out[pfx + ".norm2.bias"] = (cout,)
        out[pfx + ".conv2.weight"] = (cout, cout, 3, 3); out[pfx + ".conv2.bias"] = (cout,)
        if cin != cout:
            out[pfx + ".conv_shortcut.weight"] = (cout, cin, 1, 1); out[pfx + ".conv_shortcut.bias"] = (cout,)

    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    out[a + ".group_norm.weight"] = (top,); out[a + ".group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out[f"{a}.{n}.weight"] = (top, top); out[f"{a}.{n}.bias"] = (top,)
    resnet("decoder.mid_block.resnets.1", top, top)
    cur = top
    rev = boc[::-1]
    for i, cout in enumerate(rev):
        for j in range(c["layers_per_block"] + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", cur, cout)
            cur = cout
        if i + 1 < len(rev):
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (cout,)
    out["decoder.conv_norm_out.weight"] = (boc[0],); out["decoder.conv_norm_out.bias"] = (boc[0],)
    out["decoder.conv_out.weight"] = (c["out_channels"], boc[0], 3, 3); out["decoder.conv_out.bias"] = (c["out_channels"],)
    return out


def encoder_state_dict_shapes(cfg: dict | None = None) -> dict:
    """diffusers key -> shape of the encoder half (+ quant_conv).  Encoder (diffusers-0.29.0): conv_in Conv2d(3, 128, 3, pad 1)
    -> 4 x DownEncoderBlock2D (layers_per_block resnets; Downsample2D(padding=0): F.pad (0,1,0,1) + conv3x3 stride 2 on all
    but the last) -> UNetMidBlock2D -> GroupNorm(32, 512, 1e-6) -> SiLU -> conv_out Conv2d(512, 2L, 3, pad 1); quant_conv
    Conv2d(2L, 2L, 1)."""
    c = dict(SD14_VAE)
    if cfg:
        c.update(cfg)
    boc, L = list(c["block_out_channels"]), c["latent_channels"]
    out = {"quant_conv.weight": (2 * L, 2 * L, 1, 1), "quant_conv.bias": (2 * L,),
           "encoder.conv_in.weight": (boc[0], c["out_channels"], 3, 3), "encoder.conv_in.bias": (boc[0],)}

    def resnet(pfx, cin, cout):
        out[pfx + ".norm1.weight"] = (cin,); out[pfx + ".norm1.bias"] = (cin,)
        out[pfx + ".conv1.weight"] = (cout, cin, 3, 3); out[pfx + ".conv1.bias"] = (cout,)
        out[pfx + ".norm2.weight"] = (cout,); out[pfx + ".norm2.bias"] = (cout,)
        out[pfx + ".conv2.weight"] = (cout, cout, 3, 3); out[pfx + ".conv2.bias"] = (cout,)
        if cin != cout:
            out[pfx + ".conv_shortcut.weight"] = (cout, cin, 1, 1); out[pfx + ".conv_shortcut.bias"] = (cout,)

    cur = boc[0]
    for i, cout in enumerate(boc):
        for j in range(c["layers_per_block"]):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", cur, cout)
            cur = cout
        if i + 1 < len(boc):
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
    top = boc[-1]
    resnet("encoder.mid_block.resnets.0", top, top)
    a = "encoder.mid_block.attentions.0"
    out[a + ".group_norm.weight"] = (top,); out[a + ".group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out[f"{a}.{n}.weight"] = (top, top); out[f"{a}.{n}.bias"] = (top,)
    resnet("encoder.mid_block.resnets.1", top, top)
    out["encoder.conv_norm_out.weight"] = (top,); out["encoder.conv_norm_out.bias"] = (top,)
    out["encoder.conv_out.weight"] = (2 * L, top, 3, 3); out["encoder.conv_out.bias"] = (2 * L,)
    return out


class OracleVAEDecoder:
    def __init__(self, state_dict: dict, config: dict | None = None, act_dtype=None):
        self.cfg = dict(SD14_VAE)
        if config:
            self.cfg.update(config)
        self.q_dtype = act_dtype
        self.sd = {}
        for k, v in state_dict.items():
            v = v.detach().float()
            # matrices are stored 16-bit by the engine; vectors and the tiny post_quant_conv stay f32
            if act_dtype is not None and v.dim() > 1 and "quant_conv" not in k:
                v = v.to(act_dtype).float()
            self.sd[k] = v

    def q(self, x):
        return x if self.q_dtype is None else x.to(self.q_dtype).float()

    def P(self, name):
        return self.sd[name]

    def resnet(self, pfx, x):
        g = self.cfg["norm_groups"]
        h = self.q(F.silu(F.group_norm(x, g, self.P(pfx + ".norm1.weight"), self.P(pfx + ".norm1.bias"), eps=1e-6)))
        h = self.q(F.conv2d(h, self.P(pfx + ".conv1.weight"), self.P(pfx + ".conv1.bias"), padding=1))
        h = self.q(F.silu(F.group_norm(h, g, self.P(pfx + ".norm2.weight"), self.P(pfx + ".norm2.bias"), eps=1e-6)))
        if (pfx + ".conv_shortcut.weight") in self.sd:
            sc = self.q(F.conv2d(x, self.P(pfx + ".conv_shortcut.weight"), self.P(pfx + ".conv_shortcut.bias")))
        else:
            sc = x
        return self.q(F.conv2d(h, self.P(pfx + ".conv2.weight"), self.P(pfx + ".conv2.bias"), padding=1) + sc)

    def attention(self, pfx, x):
        b, c, hh, ww = x.shape
        g = self.cfg["norm_groups"]
        h = self.q(F.group_norm(x, g, self.P(pfx + ".group_norm.weight"), self.P(pfx + ".group_norm.bias"), eps=1e-6))
        t = h.reshape(b, c, hh * ww).transpose(1, 2)                       # [B, HW, C] tokens
        qq = self.q(F.linear(t, self.P(pfx + ".to_q.weight"), self.P(pfx + ".to_q.bias")))
        kk = self.q(F.linear(t, self.P(pfx + ".to_k.weight"), self.P(pfx + ".to_k.bias")))
        vv = self.q(F.linear(t, self.P(pfx + ".to_v.weight"), self.P(pfx + ".to_v.bias")))
        p = self.q(torch.softmax(qq @ kk.transpose(1, 2) * (c ** -0.5), dim=-1))    # heads = 1; P is stored 16-bit
        a = self.q(p @ vv)
        o = F.linear(a, self.P(pfx + ".to_out.0.weight"), self.P(pfx + ".to_out.0.bias"))
        return self.q(o.transpose(1, 2).reshape(b, c, hh, ww) + x)

    def decode(self, z: torch.Tensor, latent_scale: float = 1.0) -> torch.Tensor:
        """decoder(post_quant_conv(latent_scale * z)): [B, L, S, S] fp32 -> [B, 3, 8S, 8S] fp32."""
        c = self.cfg
        x = F.conv2d(z.float() * latent_scale, self.P("post_quant_conv.weight"), self.P("post_quant_conv.bias"))
        x = self.q(F.conv2d(x, self.P("decoder.conv_in.weight"), self.P("decoder.conv_in.bias"), padding=1))
        x = self.resnet("decoder.mid_block.resnets.0", x)
        x = self.attention("decoder.mid_block.attentions.0", x)
        x = self.resnet("decoder.mid_block.resnets.1", x)
        n = len(c["block_out_channels"])
        for i in range(n):
            for j in range(c["layers_per_block"] + 1):
                x = self.resnet(f"decoder.up_blocks.{i}.resnets.{j}", x)
            if i + 1 < n:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = self.q(F.conv2d(x, self.P(f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"),
                                    self.P(f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"), padding=1))
        x = self.q(F.silu(F.group_norm(x, c["norm_groups"], self.P("decoder.conv_norm_out.weight"),
                                       self.P("decoder.conv_norm_out.bias"), eps=1e-6)))
        return F.conv2d(x, self.P("decoder.conv_out.weight"), self.P("decoder.conv_out.bias"), padding=1)

    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """StableDiffusionPipeline.decode_latents: NHWC float32 in [0, 1] (as a tensor)."""
        image = self.decode(latents, 1.0 / self.cfg["scaling_factor"])
        return (image / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float()

    @staticmethod
    def to_uint8(images01: torch.Tensor) -> torch.Tensor:
        """numpy_to_pil's conversion: (images * 255).round().astype(uint8) (numpy rounds half to even, as torch.round)."""
        return (images01 * 255).round().to(torch.uint8)


class OracleVAEEncoder(OracleVAEDecoder):
    """encode(x) -> moments [B, 2L, S, S] (mean | logvar); embed(x, noise) = the reference's embed_fn (run_nudity.py:308)."""

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        c = self.cfg
        h = self.q(F.conv2d(x.float(), self.P("encoder.conv_in.weight"), self.P("encoder.conv_in.bias"), padding=1))
        n = len(c["block_out_channels"])
        for i in range(n):
            for j in range(c["layers_per_block"]):
                h = self.resnet(f"encoder.down_blocks.{i}.resnets.{j}", h)
            if i + 1 < n:
                h = F.pad(h, (0, 1, 0, 1))
                h = self.q(F.conv2d(h, self.P(f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"),
                                    self.P(f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"), stride=2))
        h = self.resnet("encoder.mid_block.resnets.0", h)
        h = self.attention("encoder.mid_block.attentions.0", h)
        h = self.resnet("encoder.mid_block.resnets.1", h)
        h = self.q(F.silu(F.group_norm(h, c["norm_groups"], self.P("encoder.conv_norm_out.weight"),
                                       self.P("encoder.conv_norm_out.bias"), eps=1e-6)))
        m = F.conv2d(h, self.P("encoder.conv_out.weight"), self.P("encoder.conv_out.bias"), padding=1)
        return F.conv2d(m, self.P("quant_conv.weight"), self.P("quant_conv.bias"))

    def embed(self, x: torch.Tensor, noise: torch.Tensor | None) -> torch.Tensor:
        """DiagonalGaussianDistribution.sample() * scaling_factor (noise None -> mode)."""
        mean, logvar = self.encode(x).chunk(2, dim=1)
        z = mean if noise is None else mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise
        return z * self.cfg["scaling_factor"]
