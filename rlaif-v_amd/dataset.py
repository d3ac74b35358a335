"""Sample encoding for the LLaVA-1.5 DPO path: parquet row -> (rej_dict, win_dict).

Mirrors (paths relative to /root/reference):
  conv_llava_v1 prompt              muffin/conversation.py:325-335 (+ get_prompt, SeparatorStyle.TWO :54-63)
  tokenizer_image_token             muffin/train/train_utils.py:176-195
  preprocess_v1                     muffin/train/train_utils.py:265-349  (label masking per round)
  encode_multimodal_preference_sample  muffin/train/train_utils.py:198-263
  RLAIFVDataset                     muffin/data/datasets.py:27-91        (rows with a JSON `logps` column)
  DPODataset                        muffin/train/train_llava15.py:124-145
CPU-side string / integer work executed by DataLoader workers.  The tokenizer and the CLIP image processor are the
caller's (HF objects in production; none ship offline, the tests use a deterministic toy tokenizer shared with the
golden generator that drives the reference's own functions).
"""
from __future__ import annotations

import copy
import io
import json
import os
from typing import Callable, Dict, List, Optional, Sequence

import torch

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200

SYSTEM_V1 = ("A chat between a curious human and an artificial intelligence assistant. "
             "The assistant gives helpful, detailed, and polite answers to the human's questions.")
ROLES_V1 = ("USER", "ASSISTANT")
SEP_V1, SEP2_V1 = " ", "</s>"


def llava_v1_prompt(turns: Sequence[Dict[str, str]]) -> str:
    """conv_llava_v1.get_prompt(): ``SYSTEM USER: q ASSISTANT: a</s>...`` (two-separator style)."""
    role_of = {"human": ROLES_V1[0], "gpt": ROLES_V1[1]}
    if role_of[turns[0]["from"]] != ROLES_V1[0]:
        turns = turns[1:]                                    # skip a leading non-human turn (train_utils.py:277-280)
    out = SYSTEM_V1 + SEP_V1
    for j, t in enumerate(turns):
        role = role_of[t["from"]]
        assert role == ROLES_V1[j % 2], "turns must alternate human / gpt"
        out += (role + ": " + t["value"] + (SEP_V1, SEP2_V1)[j % 2]) if t["value"] else (role + ":")
    return out


def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX, return_tensors=None):
    """Tokenise the text around every ``<image>`` tag and join the chunks with the -200 placeholder (a BOS produced
    by the tokenizer is kept once, at the front)."""
    chunks = [tokenizer(c).input_ids for c in prompt.split("<image>")]
    ids: List[int] = []
    has_bos = len(chunks) > 0 and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id
    if has_bos:
        ids.append(chunks[0][0])
    skip = 1 if has_bos else 0
    for n, c in enumerate(chunks):
        if n > 0:
            ids.append(image_token_index)
        ids.extend(c[skip:])
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    if return_tensors is not None:
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return ids


def preprocess_v1(sources: Sequence[Sequence[Dict[str, str]]], tokenizer, has_image: bool = False,
                  tokenizer_ge_0_14: bool = True) -> Dict[str, torch.Tensor]:
    prompts = [llava_v1_prompt(src) for src in sources]
    if has_image:
        input_ids = torch.stack([tokenizer_image_token(p, tokenizer, return_tensors="pt") for p in prompts], dim=0)
    else:
        input_ids = tokenizer(prompts, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length,
                              truncation=True).input_ids
    targets = input_ids.clone()
    sep = SEP_V1 + ROLES_V1[1] + ": "

    def n_tok(text: str) -> int:
        return len(tokenizer_image_token(text, tokenizer)) if has_image else len(tokenizer(text).input_ids)

    for prompt, target in zip(prompts, targets):
        total_len = int(target.ne(tokenizer.pad_token_id).sum())
        cur = 1
        target[:cur] = IGNORE_INDEX                           # BOS
        for i, rou in enumerate(prompt.split(SEP2_V1)):
            if rou == "":
                break
            parts = rou.split(sep)
            if len(parts) != 2:
                break
            round_len = n_tok(rou)
            instr_len = n_tok(parts[0] + sep) - 2
            if i != 0 and not tokenizer.legacy and tokenizer_ge_0_14:
                round_len -= 1
                instr_len -= 1
            target[cur:cur + instr_len] = IGNORE_INDEX       # system + question + "ASSISTANT:" are not targets
            cur += round_len
        target[cur:] = IGNORE_INDEX
        if cur < tokenizer.model_max_length and cur != total_len:
            target[:] = IGNORE_INDEX
            print(f"WARNING: tokenization mismatch: {cur} vs. {total_len}. (ignored)")
    return dict(input_ids=input_ids, labels=targets)


def encode_multimodal_preference_sample(source: Dict, tokenizer, multimodal_cfg: Dict,
                                        preprocess_func: Optional[Callable] = None):
    if isinstance(source["chosen"], list):
        win_conv, rej_conv = source["chosen"], source["rejected"]
    else:
        win_conv = copy.deepcopy([source["question"], source["chosen"]])
        rej_conv = copy.deepcopy([source["question"], source["rejected"]])
    image = None
    if "image" in source:
        image = multimodal_cfg["image_processor"](source["image"])
        if not multimodal_cfg.get("keep_image_tag", False):
            raise NotImplementedError("the LLaVA-1.5 path keeps the <image> tag (train_llava15.py:135)")
    preprocess_func = preprocess_func or (lambda s, t: preprocess_v1(s, t, has_image=True))
    rej = preprocess_func([rej_conv], tokenizer)
    win = preprocess_func([win_conv], tokenizer)
    rej_d = dict(input_ids=rej["input_ids"][0], labels=rej["labels"][0])
    win_d = dict(input_ids=win["input_ids"][0], labels=win["labels"][0])
    if image is not None:
        rej_d["image"] = win_d["image"] = image
    elif multimodal_cfg.get("is_multimodal", False):
        cs = multimodal_cfg["image_processor"].crop_size
        rej_d["image"] = win_d["image"] = torch.zeros(3, cs["height"], cs["width"])
    if "ref_win_logp" in source:
        for tag, d in (("rej", rej_d), ("win", win_d)):
            for k in ("logp", "avg_logp", "per_token_logp"):
                d[f"ref_{tag}_{k}"] = source[f"ref_{tag}_{k}"]
    return rej_d, win_d


def bytes_to_PIL_image(img_buffer: bytes):
    from PIL import Image
    return Image.open(io.BytesIO(img_buffer)).convert("RGB")


class RLAIFVDataset(torch.utils.data.Dataset):
    """Rows of the ``*logp*.parquet`` files under ``data_dir`` (written by inference_logp.write_logp_to_preference_parquet)."""

    def __init__(self, data_dir: str, reference_model=None, tokenizer=None, image_token_len=None, img_processor=None,
                 use_im_start_end: bool = False, is_llava15: bool = True):
        import pandas as pd
        os.makedirs(data_dir, exist_ok=True)

        def logp_files():
            return sorted(f for f in os.listdir(data_dir) if f.endswith(".parquet") and "logp" in f)

        if not logp_files():
            # muffin/data/datasets.py:38-50: no cached reference log-probs -> compute them with the reference model.  The
            # reference downloads openbmb/RLAIF-V-Dataset from the hub; here the raw rows are the *.parquet files (without
            # 'logp' in the name) found under data_dir or ./RLAIF-V-Dataset (there is no network).
            assert reference_model is not None, "`reference_model` is mandatory when logps do not exist."
            raw = [os.path.join(d, f) for d in (data_dir, "./RLAIF-V-Dataset") if os.path.isdir(d)
                   for f in sorted(os.listdir(d)) if f.endswith(".parquet") and "logp" not in f]
            if not raw:
                raise FileNotFoundError(f"no *logp*.parquet and no raw preference *.parquet under {data_dir} or ./RLAIF-V-Dataset "
                                        "(the reference would download openbmb/RLAIF-V-Dataset here; there is no network)")
            from .inference_logp import PreferenceInferenceDataset, inference_logp
            rows = pd.concat([pd.read_parquet(f) for f in raw], ignore_index=True).to_dict("records")
            inference_logp(reference_model, tokenizer,
                           PreferenceInferenceDataset(rows, tokenizer, image_token_len, img_processor, use_im_start_end),
                           data_dir, is_llava15=is_llava15)
        files = logp_files()
        self.data = pd.concat([pd.read_parquet(os.path.join(data_dir, f)) for f in files], ignore_index=True).to_dict("records")

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        s = self.data[index]
        d = {
            "image": bytes_to_PIL_image(s["image"]["bytes"]),
            "question": {"from": "human", "value": f"<image>\n{s['question']}"},
            "chosen": {"from": "gpt", "value": s["chosen"]},
            "rejected": {"from": "gpt", "value": s["rejected"]},
            "idx": s["idx"],
            "metainfo": {"origin_dataset": s.get("origin_dataset"), "origin_split": s.get("origin_split"),
                         "origin_idx": s["idx"], "image_id": s.get("image_path")},
        }
        logps = json.loads(s["logps"])
        if not isinstance(logps, list):
            logps = logps["logps"]
        (d["ref_win_logp"], d["ref_win_avg_logp"], d["ref_win_per_token_logp"],
         d["ref_rej_logp"], d["ref_rej_avg_logp"], d["ref_rej_per_token_logp"]) = logps
        return d


class DPODataset(torch.utils.data.Dataset):
    def __init__(self, tokenizer, data_dir: str, multimodal_cfg: Dict, reference_model=None):
        self.tokenizer = tokenizer
        self.list_data_dict = RLAIFVDataset(data_dir, reference_model, tokenizer, multimodal_cfg.get("image_token_len"),
                                            multimodal_cfg.get("image_processor"), multimodal_cfg.get("use_im_start_end", False),
                                            is_llava15=True)
        self.multimodal_cfg = dict(multimodal_cfg, keep_image_tag=True)

    def __len__(self):
        return len(self.list_data_dict)

    def __getitem__(self, i):
        return encode_multimodal_preference_sample(self.list_data_dict[i], self.tokenizer, self.multimodal_cfg,
                                                   preprocess_func=lambda s, t: preprocess_v1(s, t, has_image=True))
