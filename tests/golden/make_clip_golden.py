#!/usr/bin/env python3
"""Generates tests/golden/clip_golden.npz: one small randomly initialised transformers.CLIPTextModel (the third-party
module the reference imports), its state_dict, inputs and last_hidden_state with and without a padding mask.
Run in the build container (transformers 5.x is installed there): python tests/golden/make_clip_golden.py"""
import os

import numpy as np
import torch
from transformers import CLIPTextConfig, CLIPTextModel

CFG = dict(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
           max_position_embeddings=77)
torch.manual_seed(0)
m = CLIPTextModel(CLIPTextConfig(bos_token_id=510, eos_token_id=511, pad_token_id=0, **CFG)).eval()
ids = torch.randint(1, 500, (3, 77))
ids[:, 0] = 510
ids[0, 20:] = 511; ids[1, 50:] = 511; ids[2, 76] = 511
mask = (torch.arange(77)[None] <= torch.tensor([20, 50, 76])[:, None]).long()
with torch.no_grad():
    plain = m(ids)[0]
    masked = m(ids, attention_mask=mask)[0]
out = {"sd/" + k: v.numpy() for k, v in m.state_dict().items()}
out.update(ids=ids.numpy(), mask=mask.numpy(), plain=plain.numpy(), masked=masked.numpy(), cfg=np.array(sorted(CFG.items()), dtype=object))
np.savez_compressed(os.path.join(os.path.dirname(__file__), "clip_golden.npz"), **out)
print("wrote clip_golden.npz", plain.shape, float(plain.abs().mean()), float((plain - masked).abs().max()))
