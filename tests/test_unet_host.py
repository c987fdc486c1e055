"""CPU-side checks of the UNet plan builder: the manifest must reproduce public facts about SD-v1.4's
UNet2DConditionModel (686 state_dict keys, 859,520,964 parameters) and SURVEY.md's FLOP figures, and packing
must place every tensor where the manifest says."""
import math

import torch

from oracle.unet import OracleUNet
from safe_denoiser_amd.unet import P_CONV3X3, P_GEGLU_MAT, UNet2DConditionModel, _interleave16


def test_manifest_matches_public_architecture_facts():
    u = UNet2DConditionModel()
    shapes = u.state_dict_shapes()
    assert len(shapes) == 686
    assert sum(math.prod(s) for s in shapes.values()) == 859_520_964
    assert shapes["conv_in.weight"] == (320, 4, 3, 3) and shapes["conv_out.weight"] == (4, 320, 3, 3)
    assert shapes["up_blocks.1.resnets.2.conv1.weight"] == (1280, 1920, 3, 3)
    assert shapes["up_blocks.3.resnets.0.conv_shortcut.weight"] == (320, 960, 1, 1)
    assert shapes["down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight"] == (320, 768)
    assert shapes["down_blocks.2.attentions.1.transformer_blocks.0.ff.net.0.proj.weight"] == (10240, 1280)
    assert "down_blocks.3.attentions.0.norm.weight" not in shapes            # DownBlock2D has no attention
    total, attn = u.flops(1)
    assert abs(total / 1e9 - 803.3) < 0.1 and abs(attn / 1e9 - 126.1) < 0.1
    t2, a2 = u.flops(2)
    assert abs(t2 - 2 * total) / total < 1e-3                                # only the tiny time-embed GEMMs are per-call


def test_offsets_are_disjoint_and_aligned():
    u = UNet2DConditionModel()
    spans = []
    for p in u.manifest:
        esz = 4 if p["kind"] in (0, 4) else 2
        n = p["rows_padded"] * max(p["cols"], 1) * esz
        assert p["offset"] % 256 == 0
        spans.append((p["offset"], p["offset"] + n))
    spans.sort()
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0
    assert spans[-1][1] <= u.weight_bytes


def test_pack_layouts_small_config():
    u = UNet2DConditionModel(block_out_channels=(64, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                             layers_per_block=1, attention_head_dim=1, cross_attention_dim=64, sample_size=8,
                             norm_num_groups=32, text_len=5)
    sd = u.synthetic_state_dict(0)
    buf = u.pack_state_dict(sd)
    for p in u.manifest:
        t = sd[p["name"]].float()
        if p["kind"] == P_CONV3X3:
            exp = t.permute(0, 2, 3, 1).reshape(p["rows"], -1).to(torch.bfloat16)
        elif p["kind"] == P_GEGLU_MAT:
            exp = _interleave16(t).to(torch.bfloat16)
        elif p["kind"] == 1:
            exp = t.reshape(p["rows"], p["cols"]).to(torch.bfloat16)
        elif p["kind"] == 4:
            exp = _interleave16(t)
        else:
            exp = t
        raw = buf[p["offset"]:p["offset"] + exp.numel() * exp.element_size()].view(exp.dtype).reshape(exp.shape)
        assert torch.equal(raw, exp), p["name"]
    # stacked q|k|v must be byte-contiguous
    by = {p["name"]: p for p in u.manifest}
    q = by["down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight"]
    k = by["down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_k.weight"]
    assert k["offset"] == q["offset"] + q["rows"] * q["cols"] * 2
    # the oracle accepts exactly this state dict
    o = OracleUNet(sd, dict(block_out_channels=(64, 64), level_has_attn=(True, False), layers_per_block=1, n_heads=1,
                            cross_dim=64, sample_size=8))
    y = o(torch.randn(1, 4, 8, 8), 500.0, torch.randn(1, 5, 64))
    assert y.shape == (1, 4, 8, 8) and torch.isfinite(y).all()


def test_interleave16():
    t = torch.arange(64).float()
    o = _interleave16(t)
    assert o[:16].tolist() == list(range(16)) and o[16:32].tolist() == list(range(32, 48))
    assert o[32:48].tolist() == list(range(16, 32))


def test_mmdit_manifest_matches_sd3_medium_parameter_count():
    """SD3-medium's transformer has 2,028,328,000 parameters (public figure; pos_embed is a buffer, not a parameter)."""
    from safe_denoiser_amd.mmdit import SD3Transformer2DModel
    m = SD3Transformer2DModel()
    shapes = m.state_dict_shapes()
    assert sum(math.prod(s) for n, s in shapes.items() if n != "pos_embed.pos_embed") == 2_028_328_000
    assert shapes["pos_embed.pos_embed"] == (1, 192 * 192, 1536)
    assert shapes["transformer_blocks.0.norm1.linear.weight"] == (9216, 1536)
    assert shapes["transformer_blocks.23.norm1_context.linear.weight"] == (3072, 1536)      # context_pre_only
    assert "transformer_blocks.23.ff_context.net.2.weight" not in shapes
    assert shapes["proj_out.weight"] == (64, 1536) and shapes["context_embedder.weight"] == (1536, 4096)
