"""A stand-in for transformers' CLIPTokenizer in tests (its vocabulary files are not on the box): whitespace words ->
deterministic ids, BOS / EOS framing, padding with the end-of-text id (CLIP's pad token IS <|endoftext|>), the call
options the reference's pipelines use (padding="max_length" | "longest", max_length, truncation, return_tensors="pt")
and `.input_ids` / `.attention_mask` / `model_max_length`.  The end-of-text id is the LARGEST id, which is what the
pooled output's argmax relies on."""
import zlib
from types import SimpleNamespace

import torch


class FakeCLIPTokenizer:
    def __init__(self, vocab_size=49408, model_max_length=77):
        self.vocab_size, self.model_max_length = vocab_size, model_max_length
        self.bos_token_id, self.eos_token_id = vocab_size - 2, vocab_size - 1
        self.pad_token_id = self.eos_token_id

    def _ids(self, text):
        words = text.lower().replace(",", " , ").split()
        return [1 + zlib.crc32(w.encode()) % (self.vocab_size - 3) for w in words]      # 1 .. vocab-3 (0 = the mask id)

    def __call__(self, texts, padding="max_length", max_length=None, truncation=False, return_tensors="pt"):
        texts = [texts] if isinstance(texts, str) else list(texts)
        rows = [[self.bos_token_id] + self._ids(t) + [self.eos_token_id] for t in texts]
        if padding == "max_length":
            n = max_length or self.model_max_length
            if truncation:
                rows = [r if len(r) <= n else r[:n - 1] + [self.eos_token_id] for r in rows]
        else:                                                                           # "longest"
            n = max(len(r) for r in rows)
        ids = torch.full((len(rows), n), self.pad_token_id, dtype=torch.int64)
        mask = torch.zeros((len(rows), n), dtype=torch.int64)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r[:n])
            mask[i, :min(len(r), n)] = 1
        return SimpleNamespace(input_ids=ids, attention_mask=mask)
