"""Deterministic toy tokenizer with the attributes the reference's preprocess_v1 needs (llama-style: BOS first, one id
per whitespace-separated piece, '</s>' glued to the previous word becomes its own token).  Shared by the golden
generator (driving the reference) and the tests (driving the port)."""
import re
import types

import torch


class ToyTokenizer:
    bos_token_id, eos_token_id, pad_token_id, unk_token_id = 1, 2, 0, 0
    legacy = True
    model_max_length = 512

    def _ids(self, text):
        ids = [self.bos_token_id]
        for piece in re.findall(r"</s>|[^\s<]+|<", text):
            if piece == "</s>":
                ids.append(self.eos_token_id)
            else:
                ids.append(3 + (sum(ord(c) * (i + 1) for i, c in enumerate(piece)) % 480))
        return ids

    def __call__(self, text, return_tensors=None, padding=None, max_length=None, truncation=None):
        if isinstance(text, str):
            return types.SimpleNamespace(input_ids=self._ids(text))
        rows = [self._ids(t) for t in text]
        n = max(len(r) for r in rows)
        return types.SimpleNamespace(input_ids=torch.tensor([r + [self.pad_token_id] * (n - len(r)) for r in rows]))


SAMPLES = [
    dict(question="What is shown in the picture ?", chosen="A dog runs on the grass .", rejected="A cat sleeps on a sofa ."),
    dict(question="Describe the scene in detail .", chosen="Two people walk along the beach at sunset , holding hands .",
         rejected="Two people walk along the beach ."),
    dict(question="How many apples ?", chosen="Three .", rejected="There are three red apples on the table ."),
]
