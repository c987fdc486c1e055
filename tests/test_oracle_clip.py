"""CLIP text encoder row, host side: the oracle is PINNED against transformers' CLIPTextModel (the
third-party module the reference imports) through the committed golden case and, where transformers is importable, a
live second case; the engine's manifest must equal the oracle's key table."""
import math
import os

import numpy as np
import pytest
import torch

from oracle.clip import OracleCLIPText, clip_state_dict_shapes

GOLD = os.path.join(os.path.dirname(__file__), "golden", "clip_golden.npz")


def load_gold():
    z = np.load(GOLD, allow_pickle=True)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    cfg = {k: int(v) for k, v in z["cfg"]}
    return sd, cfg, torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"]), torch.from_numpy(z["plain"]), torch.from_numpy(z["masked"])


def test_oracle_matches_transformers_golden_case():
    sd, cfg, ids, mask, plain, masked = load_gold()
    o = OracleCLIPText(sd, cfg)
    torch.testing.assert_close(o(ids), plain, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(o(ids, mask), masked, rtol=2e-5, atol=2e-5)
    assert float((plain - masked).abs().max()) > 0.1                 # the padding mask is not a no-op in this case


def test_oracle_matches_live_transformers_when_available():
    tr = pytest.importorskip("transformers")
    cfg = dict(vocab_size=300, hidden_size=192, intermediate_size=384, num_hidden_layers=3, num_attention_heads=3,
               max_position_embeddings=77)
    torch.manual_seed(5)
    m = tr.CLIPTextModel(tr.CLIPTextConfig(bos_token_id=298, eos_token_id=299, pad_token_id=0, **cfg)).eval()
    ids = torch.randint(1, 298, (2, 77)); ids[:, -1] = 299
    with torch.no_grad():
        want = m(ids)[0]
    torch.testing.assert_close(OracleCLIPText(m.state_dict(), cfg)(ids), want, rtol=2e-5, atol=2e-5)


def test_full_size_key_table_and_manifest():
    shapes = clip_state_dict_shapes()
    assert sum(math.prod(s) for s in shapes.values()) == 123_060_480        # CLIP ViT-L/14 text model (public figure)
    assert len(shapes) == 2 + 12 * 16 + 2
    pytest.importorskip("safe_denoiser_amd")
    from safe_denoiser_amd.clip import CLIPTextModel
    try:
        m = CLIPTextModel()
    except Exception as e:
        pytest.skip(str(e))
    assert m.state_dict_shapes() == shapes
    assert set(CLIPTextModel._canonical({"text_model." + k: 0 for k in shapes})) == set(shapes)
