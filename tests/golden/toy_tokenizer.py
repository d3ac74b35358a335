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


class OmniToyTokenizer:
    """Toy tokenizer with the surface omni_preprocess needs (omnilmm/train/train_utils.py:50-151): a Zephyr-style chat
    template, ``encode(add_special_tokens=False)``, ``__call__(..., return_tensors="pt")`` with BOS, ``decode``.  Newlines, role
    markers and the image tokens are their own pieces, so a marker tokenizes the same inside a text and alone."""
    bos_token_id, eos_token_id, pad_token_id, unk_token_id = 1, 2, 0, 0
    model_max_length = 96
    SPECIAL = {"</s>": 2, "\n": 3, "<|system|>": 4, "<|user|>": 5, "<|assistant|>": 6, "<im_start>": 7, "<im_end>": 8,
               "<im_patch>": 9}

    def _pieces(self, text):
        return re.findall(r"</s>|<\|[a-z]+\|>|<im_start>|<im_end>|<im_patch>|\n|[^\s<]+|<", text)

    def encode(self, text, add_special_tokens=True):
        ids = [self.bos_token_id] if add_special_tokens else []
        for p in self._pieces(text):
            ids.append(self.SPECIAL[p] if p in self.SPECIAL else 10 + (sum(ord(c) * (i + 1) for i, c in enumerate(p)) % 470))
        return ids

    def __call__(self, text, return_tensors=None, padding=None, max_length=None, truncation=None):
        ids = self.encode(text)
        if truncation and max_length is not None:
            ids = ids[:max_length]
        return types.SimpleNamespace(input_ids=torch.tensor([ids]))

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=False):
        text = "".join(f"<|{m['role']}|>\n{m['content']}</s>\n" for m in messages)
        return text + ("<|assistant|>\n" if add_generation_prompt else "")


OMNI_CONVERSATIONS = [
    [{"from": "human", "value": "<im_start><im_patch><im_patch><im_end>\nWhat is shown in the picture ?"},
     {"from": "gpt", "value": "A dog runs on the grass ."}],
    [{"role": "user", "content": "Describe the scene ."}, {"role": "assistant", "content": "Two people walk along the beach ."},
     {"role": "user", "content": "And the weather ?"}, {"role": "assistant", "content": "Sunny , with a few clouds ."}],
    [{"role": "user", "content": "How many apples ?"}, {"role": "assistant", "content": "Three ."},
     {"role": "user", "content": "Are you sure ?"}],                                     # trailing question without an answer
    [{"role": "user", "content": " ".join(f"word{i}" for i in range(120))},             # truncated before any assistant marker
     {"role": "assistant", "content": "never reached"}],
]
