"""CPU oracle: CLIP text encoder, plain torch ops.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

The live module is transformers' CLIPTextModel (third party; the reference imports it and calls
`self.text_encoder(input_ids, attention_mask=...)[0]` at
models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:197,225,287,333, and re-states its causal mask
at :126-134).  PARITY PINNED against that third-party implementation: `transformers` IS installed in the build image, so
tests/test_oracle_clip.py runs a randomly initialised transformers.CLIPTextModel beside this restatement on the same
weights (fp32, tolerance 2e-5) and tests/golden/clip_golden.npz stores one such case for machines without it.

Definition restated: x = tok_emb[ids] + pos_emb; for each layer: h = LN1(x); q,k,v = linear(h); causal (+ key padding)
softmax(q k^T / 8) v over 12 heads of 64; x = x + out_proj(.); h = LN2(x); x = x + fc2(quick_gelu(fc1(h))),
quick_gelu(u) = u * sigmoid(1.702 u); output = final_layer_norm(x).  All LayerNorm eps 1e-5.
``act_dtype`` emulates the engine's 16-bit storage points; None = pure fp32.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SD14_CLIP = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                 max_position_embeddings=77)


def clip_state_dict_shapes(cfg: dict | None = None) -> dict:
    c = dict(SD14_CLIP)
    if cfg:
        c.update(cfg)
    C_, I = c["hidden_size"], c["intermediate_size"]
    out = {"embeddings.token_embedding.weight": (c["vocab_size"], C_),
           "embeddings.position_embedding.weight": (c["max_position_embeddings"], C_)}
    for l in range(c["num_hidden_layers"]):
        p = f"encoder.layers.{l}"
        for n in ("layer_norm1", "layer_norm2"):
            out[f"{p}.{n}.weight"] = (C_,); out[f"{p}.{n}.bias"] = (C_,)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[f"{p}.self_attn.{n}.weight"] = (C_, C_); out[f"{p}.self_attn.{n}.bias"] = (C_,)
        out[f"{p}.mlp.fc1.weight"] = (I, C_); out[f"{p}.mlp.fc1.bias"] = (I,)
        out[f"{p}.mlp.fc2.weight"] = (C_, I); out[f"{p}.mlp.fc2.bias"] = (C_,)
    out["final_layer_norm.weight"] = (C_,); out["final_layer_norm.bias"] = (C_,)
    return out


class OracleCLIPText:
    def __init__(self, state_dict: dict, config: dict | None = None, act_dtype=None, device=None):
        """`device`: evaluate the same torch ops there (a full-size encoder over ~100 sequences is minutes on a few host cores);
        the arithmetic is unchanged fp32 (callers switch TF32 off)."""
        self.device = device
        self.cfg = dict(SD14_CLIP)
        if config:
            self.cfg.update(config)
        self.q_dtype = act_dtype
        self.sd = {}
        for k, v in state_dict.items():
            k = k[len("text_model."):] if k.startswith("text_model.") else k
            v = v.detach().float()
            if act_dtype is not None and v.dim() > 1:
                v = v.to(act_dtype).float()
            self.sd[k] = v if device is None else v.to(device)

    def q(self, x):
        return x if self.q_dtype is None else x.to(self.q_dtype).float()

    def __call__(self, input_ids: torch.Tensor, attention_mask: torch.Tensor | None = None) -> torch.Tensor:
        c, P = self.cfg, self.sd
        if self.device is not None:
            input_ids = input_ids.to(self.device)
            attention_mask = None if attention_mask is None else attention_mask.to(self.device)
        b, n = input_ids.shape
        C_, H = c["hidden_size"], c["num_attention_heads"]
        d = C_ // H
        x = self.q(P["embeddings.token_embedding.weight"][input_ids.long()] + P["embeddings.position_embedding.weight"][:n][None])
        bias = torch.full((n, n), float("-inf"), device=input_ids.device).triu_(1)[None, None]   # causal: key <= query
        if attention_mask is not None:
            bias = bias + torch.where(attention_mask[:, None, None, :] != 0, 0.0, float("-inf"))
        for l in range(c["num_hidden_layers"]):
            p = f"encoder.layers.{l}"
            h = self.q(F.layer_norm(x, (C_,), P[p + ".layer_norm1.weight"], P[p + ".layer_norm1.bias"], 1e-5))
            qq, kk, vv = (self.q(F.linear(h, P[f"{p}.self_attn.{t}_proj.weight"], P[f"{p}.self_attn.{t}_proj.bias"]))
                          .reshape(b, n, H, d).transpose(1, 2) for t in ("q", "k", "v"))
            a = torch.softmax(qq @ kk.transpose(-1, -2) * d ** -0.5 + bias, dim=-1) @ vv
            a = self.q(a.transpose(1, 2).reshape(b, n, C_))
            x = self.q(x + F.linear(a, P[p + ".self_attn.out_proj.weight"], P[p + ".self_attn.out_proj.bias"]))
            h = self.q(F.layer_norm(x, (C_,), P[p + ".layer_norm2.weight"], P[p + ".layer_norm2.bias"], 1e-5))
            u = F.linear(h, P[p + ".mlp.fc1.weight"], P[p + ".mlp.fc1.bias"])
            u = self.q(u * torch.sigmoid(1.702 * u))
            x = self.q(x + F.linear(u, P[p + ".mlp.fc2.weight"], P[p + ".mlp.fc2.bias"]))
        return self.q(F.layer_norm(x, (C_,), P["final_layer_norm.weight"], P["final_layer_norm.bias"], 1e-5))
