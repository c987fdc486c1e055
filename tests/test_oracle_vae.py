"""Host-side checks of the VAE decoder row: the oracle's wiring against public facts about the
architecture, and the engine's parameter manifest against the oracle's key/shape table (no GPU needed: the manifest is
host logic of libsdn)."""
import math

import pytest
import torch

from oracle.vae import OracleVAEDecoder, OracleVAEEncoder, decoder_state_dict_shapes, encoder_state_dict_shapes


def test_decoder_parameter_count_matches_published_architecture():
    shapes = decoder_state_dict_shapes()
    dec = sum(math.prod(s) for k, s in shapes.items() if k.startswith("decoder."))
    pq = sum(math.prod(s) for k, s in shapes.items() if k.startswith("post_quant_conv."))
    assert dec == 49_490_179 and pq == 20            # AutoencoderKL total 83,653,863 = 34,163,592 + 49,490,179 + 72 + 20
    assert len(shapes) == 138 + 2


def test_encoder_parameter_count_matches_published_architecture():
    shapes = encoder_state_dict_shapes()
    enc = sum(math.prod(s) for k, s in shapes.items() if k.startswith("encoder."))
    qc = sum(math.prod(s) for k, s in shapes.items() if k.startswith("quant_conv."))
    assert enc == 34_163_592 and qc == 72


def test_tiny_encoder_halves_resolution_per_level_and_pads_right_bottom():
    cfg = dict(block_out_channels=(64, 128), layers_per_block=1)
    shapes = encoder_state_dict_shapes(cfg)
    g = torch.Generator().manual_seed(0)
    sd = {k: torch.randn(s, generator=g) * (0.05 if len(s) > 1 else 0.1) + (1.0 if "norm" in k and k.endswith("weight") else 0.0)
          for k, s in shapes.items()}
    o = OracleVAEEncoder(sd, cfg)
    x = torch.randn(2, 3, 16, 16, generator=g)
    m = o.encode(x)
    assert m.shape == (2, 8, 8, 8) and torch.isfinite(m).all()
    z0 = o.embed(x, None)
    torch.testing.assert_close(z0, m[:, :4] * 0.18215)
    n = torch.randn(2, 4, 8, 8, generator=g)
    torch.testing.assert_close(o.embed(x, n), (m[:, :4] + torch.exp(0.5 * m[:, 4:].clamp(-30, 20)) * n) * 0.18215)


def test_tiny_decoder_runs_and_upsamples_by_2_pow_levels_minus_1():
    cfg = dict(block_out_channels=(64, 128), layers_per_block=1, norm_groups=32)
    shapes = decoder_state_dict_shapes(cfg)
    g = torch.Generator().manual_seed(0)
    sd = {k: torch.randn(s, generator=g) * (0.05 if len(s) > 1 else 0.1) + (1.0 if k.endswith("norm1.weight") else 0.0)
          for k, s in shapes.items()}
    o = OracleVAEDecoder(sd, cfg)
    z = torch.randn(2, 4, 8, 8, generator=g)
    img = o.decode(z)
    assert img.shape == (2, 3, 16, 16) and torch.isfinite(img).all()
    # batch rows independent; latent_scale is a plain pre-multiplication
    torch.testing.assert_close(o.decode(z[:1]), img[:1])
    torch.testing.assert_close(o.decode(z, 2.0), o.decode(2.0 * z))
    im01 = o.decode_latents(z)
    assert im01.shape == (2, 16, 16, 3) and float(im01.min()) >= 0.0 and float(im01.max()) <= 1.0
    u8 = o.to_uint8(torch.tensor([0.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 1.0]))
    assert u8.tolist() == [0, 0, 2, 2, 255]          # round half to even, as numpy


def test_engine_manifest_equals_oracle_key_table():
    pytest.importorskip("safe_denoiser_amd")
    from safe_denoiser_amd.vae import AutoencoderKL
    try:
        v = AutoencoderKL()
    except Exception as e:                               # library not built in this checkout
        pytest.skip(str(e))
    shapes = decoder_state_dict_shapes()
    assert v.state_dict_shapes() == shapes
    n = sum(p["rows"] * max(p["cols"], 1) for p in v.manifest)
    assert n == 49_490_179 + 20
    # deprecated on-disk attention names are aliases
    sd = {k.replace("to_q", "query").replace("to_k", "key").replace("to_v", "value").replace("to_out.0", "proj_attn"): torch.zeros(s)
          for k, s in shapes.items()}
    assert set(AutoencoderKL._canonical(sd)) == set(shapes)
    e = v._encoder()
    assert e.state_dict_shapes() == encoder_state_dict_shapes()
    assert sum(p["rows"] * max(p["cols"], 1) for p in e.manifest) == 34_163_592 + 72
