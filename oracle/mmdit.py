"""CPU oracle: SD-v3 MMDiT forward and the SD-v3 loop body (row P4), plain torch ops.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

PARITY UNPINNED at the reference level: SD3Transformer2DModel lives only in diffusers==0.29.0 (absent here; the
reference imports it at models/sdv3/safe_denoiser_pipeline.py:30 and calls it at :1120-1127) and the reference holds no
tests for it.  This restates the published diffusers-0.29.0 definitions: PatchEmbed (conv k=s=2 + centre-cropped
pos_embed), CombinedTimestepTextProjEmbeddings, JointTransformerBlock with AdaLayerNormZero (chunk order shift_msa,
scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp), AdaLayerNormContinuous for the last block's context and for
norm_out (chunk order scale, shift), JointAttnProcessor2_0 (image tokens then text tokens), FeedForward with
GELU(approximate="tanh"), proj_out + unpatchify ("nhwpqc->nchpwq").  One structural fact pins the wiring: the
SD3-medium configuration yields 2,028,328,000 transformer parameters (tests/test_unet_host.py).

`act_dtype` emulates the engine's 16-bit storage points (weights and every tensor written to HBM as 16 bit).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

SD3 = dict(in_channels=16, out_channels=16, sample_size=64, patch_size=2, num_layers=24, num_heads=24, head_dim=64,
           joint_dim=4096, pooled_dim=2048, pos_embed_max_size=192, time_dim=256)


class OracleMMDiT:
    def __init__(self, state_dict: dict, config: dict | None = None, act_dtype=None, device=None):
        """`device`: evaluate the same torch ops there (the full 2 B-parameter network over a loop is hours on a few host
        cores); arithmetic unchanged fp32 (callers switch TF32 off)."""
        self.device = device
        self.cfg = dict(SD3)
        if config:
            self.cfg.update(config)
        self.q_dtype = act_dtype
        self.sd = {}
        for k, v in state_dict.items():
            v = v.detach().float()
            if act_dtype is not None and v.dim() > 1:
                v = v.to(act_dtype).float()
            self.sd[k] = v if device is None else v.to(device)

    def q(self, x):
        return x if self.q_dtype is None else x.to(self.q_dtype).float()

    def P(self, n):
        return self.sd[n]

    def lin(self, x, pfx):
        return F.linear(x, self.P(pfx + ".weight"), self.P(pfx + ".bias"))

    @staticmethod
    def ln(x):
        return F.layer_norm(x, (x.shape[-1],), eps=1e-6)

    @torch.no_grad()
    def __call__(self, hidden_states, timestep: float, encoder_hidden_states, pooled_projections):
        c = self.cfg
        C_ = c["num_heads"] * c["head_dim"]
        b = hidden_states.shape[0]
        ps, m = c["patch_size"], c["pos_embed_max_size"]
        hp = c["sample_size"] // ps
        # PatchEmbed
        x = F.conv2d(self.q(hidden_states.float()), self.P("pos_embed.proj.weight"), self.P("pos_embed.proj.bias"), stride=ps)
        x = x.flatten(2).transpose(1, 2)                                        # [B, N, C]
        top = (m - hp) // 2
        pos = self.P("pos_embed.pos_embed").reshape(m, m, -1)[top:top + hp, top:top + hp].reshape(1, hp * hp, -1)
        x = self.q(x + pos)
        # conditioning
        half = c["time_dim"] // 2
        fr = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half).to(hidden_states.device)
        ang = torch.full((b, 1), float(timestep), device=hidden_states.device) * fr[None]
        tsin = self.q(torch.cat([torch.cos(ang), torch.sin(ang)], -1))
        te = self.q(F.silu(self.lin(tsin, "time_text_embed.timestep_embedder.linear_1")))
        te = self.lin(te, "time_text_embed.timestep_embedder.linear_2")           # f32
        pe = self.q(F.silu(self.lin(self.q(pooled_projections.float()), "time_text_embed.text_embedder.linear_1")))
        pe = self.lin(pe, "time_text_embed.text_embedder.linear_2")
        scond = self.q(F.silu(te + pe))
        ctx = self.q(self.lin(self.q(encoder_hidden_states.float()), "context_embedder"))
        L = c["num_layers"]
        for i in range(L):
            pfx = f"transformer_blocks.{i}"
            last = i == L - 1
            sh_a, sc_a, g_a, sh_m, sc_m, g_m = self.lin(scond, pfx + ".norm1.linear").chunk(6, dim=1)
            xn = self.q(self.ln(x) * (1 + sc_a[:, None]) + sh_a[:, None])
            if last:
                c_sc, c_sh = self.lin(scond, pfx + ".norm1_context.linear").chunk(2, dim=1)
                cn = self.q(self.ln(ctx) * (1 + c_sc[:, None]) + c_sh[:, None])
            else:
                c_sh_a, c_sc_a, c_g_a, c_sh_m, c_sc_m, c_g_m = self.lin(scond, pfx + ".norm1_context.linear").chunk(6, dim=1)
                cn = self.q(self.ln(ctx) * (1 + c_sc_a[:, None]) + c_sh_a[:, None])
            qx, kx, vx = (self.q(self.lin(xn, f"{pfx}.attn.to_{n}")) for n in "qkv")
            qc, kc, vc = (self.q(self.lin(cn, f"{pfx}.attn.add_{n}_proj")) for n in "qkv")
            N = x.shape[1]
            sp = lambda t: t.reshape(b, -1, c["num_heads"], c["head_dim"]).transpose(1, 2)
            a = F.scaled_dot_product_attention(sp(torch.cat([qx, qc], 1)), sp(torch.cat([kx, kc], 1)), sp(torch.cat([vx, vc], 1)))
            a = self.q(a.transpose(1, 2).reshape(b, -1, C_))
            ax, ac = a[:, :N], a[:, N:]
            x = self.q(x + g_a[:, None] * self.lin(ax, pfx + ".attn.to_out.0"))
            xm = self.q(self.ln(x) * (1 + sc_m[:, None]) + sh_m[:, None])
            h = self.q(F.gelu(self.lin(xm, pfx + ".ff.net.0.proj"), approximate="tanh"))
            x = self.q(x + g_m[:, None] * self.lin(h, pfx + ".ff.net.2"))
            if not last:
                ctx = self.q(ctx + c_g_a[:, None] * self.lin(ac, pfx + ".attn.to_add_out"))
                cm = self.q(self.ln(ctx) * (1 + c_sc_m[:, None]) + c_sh_m[:, None])
                hc = self.q(F.gelu(self.lin(cm, pfx + ".ff_context.net.0.proj"), approximate="tanh"))
                ctx = self.q(ctx + c_g_m[:, None] * self.lin(hc, pfx + ".ff_context.net.2"))
        sc, sh = self.lin(scond, "norm_out.linear").chunk(2, dim=1)
        x = self.q(self.ln(x) * (1 + sc[:, None]) + sh[:, None])
        tok = self.lin(x, "proj_out")                                               # [B, N, p*p*Cout] f32
        co = c["out_channels"]
        tok = tok.reshape(b, hp, hp, ps, ps, co)
        return torch.einsum("nhwpqc->nchpwq", tok).reshape(b, co, hp * ps, hp * ps)


def sd3_denoise_one(transformer, scheduler, embeds_pair, pooled_pair, p, noise_fn, *, num_inference_steps=50,
                    guidance_scale=7.0, repel=None, latents_dtype=torch.float16, negation_warmup_start=1000,
                    negation_warmup_end=780):
    """The reference's SD-v3 loop for ONE prompt (models/sdv3/safe_denoiser_pipeline.py:1105-1171).
    embeds_pair [2,T,4096] / pooled_pair [2,2048] = (negative, positive).  repel = dict(proj_refs=..., scale=...) uses
    the fast_sdv3 kernel_fast projection.  Noise comes from noise_fn(p, shape) in draw order (latents, then one z per
    window step)."""
    from . import repellency as orp
    from . import schedulers as osch
    cfg = transformer.cfg
    shape = (1, cfg["in_channels"], cfg["sample_size"], cfg["sample_size"])
    rq = (lambda x: x) if latents_dtype == torch.float32 else (lambda x: x.to(latents_dtype).float())
    scheduler.set_timesteps(num_inference_steps)
    ts = scheduler.timesteps.tolist()
    lat = rq(noise_fn(p, shape))
    n_win = 0
    for i, t in enumerate(ts):
        out = rq(transformer(torch.cat([lat] * 2), float(t), embeds_pair, pooled_pair))
        v = out[0:1] + guidance_scale * (out[1:2] - out[0:1])
        if negation_warmup_end <= t <= negation_warmup_start and repel is not None:
            n_win += 1
            sigma = t / 1000.0
            sigma_next = ts[i + 1] / 1000.0 if i + 1 < len(ts) else 0.0
            z = noise_fn(p, shape)
            x0r_fn = lambda x0: orp.kernel_fast_conditioning(rq(x0), repel["proj_refs"], flavour="fast_sdv3",
                                                             scale=repel["scale"])["x_0_hat"]
            # x1 is rounded to the latents dtype exactly where the engine does
            x0 = lat - sigma * v
            x1 = rq(lat + (1 - sigma) * v)
            x0r = x0r_fn(x0)
            noise = math.sqrt(sigma_next) * x1 + math.sqrt(1 - sigma_next) * z
            lat = x0r + sigma_next * (noise - x0r)
            scheduler._i += 1
        else:
            lat = scheduler.step(v, t, lat)
        lat = rq(lat)
    return lat, {"window_steps": n_win}
