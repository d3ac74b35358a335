"""OmniLMM sample encoding (BASELINE config 4): chat-template text -> ids + response-only labels.

Mirrors (paths relative to /root/reference):
  omni_preprocess                    omnilmm/train/train_utils.py:50-151
  expand_question_into_multimodal    chat.py:62-69
  wrap_question_for_omni_lmm         chat.py:71-85
CPU-side string / integer work.  The tokenizer is the caller's (an HF tokenizer with a chat template in production - Zephyr's
``<|system|> / <|user|> / <|assistant|>`` turns); the tests drive this port and the reference's own function with the same
deterministic toy tokenizer (tests/golden/toy_tokenizer.py::OmniToyTokenizer, tests/golden/make_omni_preprocess_golden.py).

Label rule (what the reference's loop over template hits amounts to): every token up to and including the first
``\\n<|assistant|>\\n`` marker is ignored; afterwards each span from a ``\\n<|user|>\\n`` marker up to the end of the next
assistant marker is ignored; a trailing user turn without an answer is ignored to the end; a text without any assistant
marker (or without any user marker) contributes nothing to the loss (a warning, all labels -100).
"""
from __future__ import annotations

import warnings
from typing import Dict, List, Sequence

import torch

IGNORE_INDEX = -100
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"

SYSTEM_CONTENT = ("You are an artificial intelligence assistant, which gives helpful, detailed, and polite answers to the "
                  "human's questions.")
RESPONSE_TEMPLATE = "\n<|assistant|>\n"
INSTRUCTION_TEMPLATE = "\n<|user|>\n"
_ROLE = {"human": "user", "gpt": "assistant", "user": "user", "assistant": "assistant"}


def _occurrences(ids: List[int], pattern: List[int]) -> List[int]:
    """Start positions of every occurrence of ``pattern`` in ``ids`` (overlaps included, like the reference's scan)."""
    n, m = len(ids), len(pattern)
    if m == 0:
        return []
    return [i for i in range(n - m + 1) if ids[i] == pattern[0] and ids[i:i + m] == pattern]


def _normalise_turns(conv: Sequence[Dict[str, str]]) -> List[Dict[str, str]]:
    turns: List[Dict[str, str]] = []
    prev = None
    for t in conv:
        role = t["from"] if "from" in t else t["role"]
        content = t["value"] if "value" in t else t["content"]
        if role not in _ROLE:
            raise AssertionError(f"role {role!r}: expected user / assistant (human / gpt)")
        role = _ROLE[role]
        assert role != prev, f"role={role}, prev_role={prev}"          # turns must alternate
        prev = role
        turns.append({"role": role, "content": content})
    if turns[0]["role"] != "system":
        turns.insert(0, {"role": "system", "content": SYSTEM_CONTENT})
    return turns


def omni_preprocess(sources: Sequence[Sequence[Dict[str, str]]], tokenizer, generation: bool = False) -> Dict[str, list]:
    """omnilmm/train/train_utils.py:50-151: one (input_ids, labels) pair per conversation; ``generation`` appends the
    generation prompt (inference) instead of stripping the text."""
    resp = list(tokenizer.encode(RESPONSE_TEMPLATE, add_special_tokens=False))
    inst = list(tokenizer.encode(INSTRUCTION_TEMPLATE, add_special_tokens=False))
    out_ids, out_labels = [], []
    for conv in sources:
        turns = _normalise_turns(conv)
        text = tokenizer.apply_chat_template(turns, tokenize=False, add_generation_prompt=generation)
        if not generation:
            text = text.strip()
        ids = tokenizer(text, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length,
                        truncation=True).input_ids[0]
        labels = ids.clone()
        flat = ids.tolist()
        answer_starts = [p + len(resp) for p in _occurrences(flat, resp)]       # first token AFTER each assistant marker
        user_starts = _occurrences(flat, inst)
        for what, hits in ((RESPONSE_TEMPLATE, answer_starts), (INSTRUCTION_TEMPLATE, user_starts)):
            if not hits:
                warnings.warn(f"Could not find key `{what}` in the following instance: @===>{tokenizer.decode(ids)}<===@ "
                              f"Raw text is @===>{text}<===@ This instance will be ignored in loss calculation. "
                              "Note, if this happens often, consider increasing the `max_seq_length`.")
                labels[:] = IGNORE_INDEX
        for k, (start, end) in enumerate(zip(user_starts, answer_starts)):
            labels[(0 if k == 0 else start):end] = IGNORE_INDEX               # everything that is not a response
        if len(answer_starts) < len(user_starts):
            labels[user_starts[-1]:] = IGNORE_INDEX                           # a trailing question without an answer
        out_ids.append(ids)
        out_labels.append(labels)
    return dict(input_ids=out_ids, labels=out_labels)


def expand_question_into_multimodal(question_text, image_token_len: int, im_st_token: str, im_ed_token: str,
                                    im_patch_token: str):
    """chat.py:62-69: the first turn's ``<image>`` tag (or, without one, the front of the turn) becomes
    ``<im_start>`` + ``<im_patch>`` x image_token_len + ``<im_end>``.  Edits the conversation in place, like the reference."""
    span = im_st_token + im_patch_token * image_token_len + im_ed_token
    first = question_text[0]
    if DEFAULT_IMAGE_TOKEN in first["content"]:
        first["content"] = first["content"].replace(DEFAULT_IMAGE_TOKEN, span)
    else:
        first["content"] = span + "\n" + first["content"]
    return question_text


def wrap_question_for_omni_lmm(question, image_token_len: int, tokenizer) -> Dict[str, torch.Tensor]:
    """chat.py:71-85: a question (string or conversation) -> the model's input ids / labels with the generation prompt."""
    if isinstance(question, str):
        question = [{"role": "user", "content": question}]
    conv = expand_question_into_multimodal(question, image_token_len, DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN,
                                           DEFAULT_IMAGE_PATCH_TOKEN)
    d = omni_preprocess(sources=[conv], tokenizer=tokenizer, generation=True)
    return dict(input_ids=d["input_ids"][0], labels=d["labels"][0])
