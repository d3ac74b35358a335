"""Batch construction for the DPO step: the reference's collator surface (same class / function names, argument meaning and
batch-dict keys).

STATUS: OFFLINE / BENCH STAND-IN.  At the integration point the reference's OWN collator
(muffin/train/train_muffin.py:37-112) is the documented default feed of the trainer: this module restates
it only because tests, bench.py and the GPU box cannot import /root/reference.  It is pinned against a fixture produced by the
reference collator (tests/test_host_logic.py::test_collator_matches_reference_golden); do not grow it - new host-side behaviour
belongs in the reference's collator, not in a twin.

Mirrors (paths relative to /root/reference):
  SFT_collator_fn            muffin/train/train_utils.py:55-96
  concate_pad, preference_collator_fn   muffin/eval/muffin_inference_logp.py:180-208
  DataCollatorForDPODataset  muffin/train/train_muffin.py:37-112
  get_diff_ids               utils/diff_lib.py:110-176   (difflib token diff for token weights)
This is CPU-side integer work done by DataLoader workers; nothing here touches the GPU.
"""
from __future__ import annotations

import difflib
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import math

import torch
from typing import Optional

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200


def SFT_collator_fn(instances, pad_token_id):
    input_ids, labels = tuple([instance[key] for instance in instances] for key in ("input_ids", "labels"))
    input_ids = torch.nn.utils.rnn.pad_sequence(input_ids, batch_first=True, padding_value=pad_token_id)
    labels = torch.nn.utils.rnn.pad_sequence(labels, batch_first=True, padding_value=IGNORE_INDEX)
    batch = dict(input_ids=input_ids, labels=labels, attention_mask=input_ids.ne(pad_token_id))
    images = [instance["image"] for instance in instances if "image" in instance]
    if len(images) > 0:
        if len(images[0].shape) == 4:
            batch["images"] = images
        elif all(x is not None and x.shape == images[0].shape for x in images):
            batch["images"] = torch.stack([torch.as_tensor(x) for x in images])
        else:
            batch["images"] = images
    else:
        batch["images"] = []
    return batch


def concate_pad(tensorA, tensorB, padding_value):
    return torch.nn.utils.rnn.pad_sequence(list(tensorA) + list(tensorB), batch_first=True,
                                           padding_value=padding_value)


def preference_collator_fn(instances, pad_token_id):
    rej_instances, win_instances = list(zip(*instances))
    rej_batch = SFT_collator_fn(rej_instances, pad_token_id)
    win_batch = SFT_collator_fn(win_instances, pad_token_id)
    concatenated_input_ids = concate_pad(win_batch["input_ids"], rej_batch["input_ids"], pad_token_id)
    concatenated_labels = concate_pad(win_batch["labels"], rej_batch["labels"], IGNORE_INDEX)
    return dict(
        concatenated_input_ids=concatenated_input_ids,
        concatenated_labels=concatenated_labels,
        concatenated_attention_mask=concatenated_input_ids.ne(pad_token_id),
        win_input_ids=win_batch["input_ids"], rej_input_ids=rej_batch["input_ids"],
        win_labels=win_batch["labels"], rej_labels=rej_batch["labels"],
        win_attention_mask=win_batch["attention_mask"], rej_attention_mask=rej_batch["attention_mask"],
        images=win_batch["images"],
    )


# ---- utils/diff_lib.py restated ------------------------------------------------------------------
def _get_match_info(a_seq, b_seq, min_match_size=1):
    mb = difflib.SequenceMatcher(None, a_seq, b_seq).get_matching_blocks()
    mb = [m for m in mb[:-1] if m[2] >= min_match_size] + [mb[-1]]
    return [(x[0], x[0] + x[2]) for x in mb], [(x[1], x[1] + x[2]) for x in mb]


def _complete_modification_spans(matches, length):
    i, j = 0, matches[0][0]
    out = []
    for idx in range(len(matches)):
        out.append((i, j))
        out.append(matches[idx])
        if idx + 1 < len(matches):
            i, j = matches[idx][1], matches[idx + 1][0]
        else:
            i, j = matches[idx][1], length
    return out


def get_diff_ids(a_seq, b_seq, min_match_size=3) -> Tuple[List[int], List[int]]:
    a_matches, b_matches = _get_match_info(a_seq, b_seq, min_match_size)
    a_spans = _complete_modification_spans(a_matches, len(a_seq))
    b_spans = _complete_modification_spans(b_matches, len(b_seq))
    mod_map = {}
    for idx, (a_span, b_span) in enumerate(zip(a_spans, b_spans)):
        if idx % 2 == 1:
            continue
        if a_span[0] != a_span[1] and b_span[0] != b_span[1]:
            mod_map[a_span] = b_span

    def spans2ids(spans):
        ids = []
        for s in spans:
            ids += list(range(s[0], s[1]))
        return sorted(set(ids))

    return spans2ids(mod_map.keys()), spans2ids(mod_map.values())


@dataclass
class DataCollatorForDPODataset(object):
    """instances: sequence of (rej_dict, win_dict) as produced by DPODataset.__getitem__
    (muffin/train/train_llava15.py:140-145)."""
    tokenizer: object          # anything with .pad_token_id
    beta: float
    mod_token_weight: float

    def __call__(self, instances: Sequence[Tuple[Dict, Dict]]) -> Dict[str, torch.Tensor]:
        batch = preference_collator_fn(instances, self.tokenizer.pad_token_id)
        rej_instances, win_instances = list(zip(*instances))
        batch["beta"] = self.beta
        batch["ref_win_logp"] = torch.as_tensor([x["ref_win_logp"] for x in win_instances])
        batch["ref_rej_logp"] = torch.as_tensor([x["ref_rej_logp"] for x in rej_instances])
        batch["ref_win_avg_logp"] = torch.as_tensor([x["ref_win_avg_logp"] for x in win_instances])
        batch["ref_rej_avg_logp"] = torch.as_tensor([x["ref_rej_avg_logp"] for x in rej_instances])
        ref_win_pt = [torch.as_tensor(x["ref_win_per_token_logp"]) for x in win_instances]
        ref_rej_pt = [torch.as_tensor(x["ref_rej_per_token_logp"]) for x in rej_instances]
        batch["ref_win_per_token_logp"] = torch.nn.utils.rnn.pad_sequence(ref_win_pt, batch_first=True, padding_value=0)
        batch["ref_rej_per_token_logp"] = torch.nn.utils.rnn.pad_sequence(ref_rej_pt, batch_first=True, padding_value=0)
        win_input_ids, rej_input_ids = batch["win_input_ids"], batch["rej_input_ids"]
        assert batch["ref_win_per_token_logp"].size(1) >= win_input_ids.size(1) - 1, \
            f"{batch['ref_win_per_token_logp'].size(1)} >= {win_input_ids.size(1) - 1}"
        assert batch["ref_rej_per_token_logp"].size(1) >= rej_input_ids.size(1) - 1, \
            f"{batch['ref_rej_per_token_logp'].size(1)} >= {rej_input_ids.size(1) - 1}"
        # one token shorter: the last position's output is never used
        batch["ref_win_per_token_logp"] = batch["ref_win_per_token_logp"][:, :win_input_ids.size(1) - 1]
        batch["ref_rej_per_token_logp"] = batch["ref_rej_per_token_logp"][:, :rej_input_ids.size(1) - 1]
        win_token_weight = torch.ones_like(batch["ref_win_per_token_logp"])
        rej_token_weight = torch.ones_like(batch["ref_rej_per_token_logp"])
        for idx, (w, r) in enumerate(zip(win_input_ids, rej_input_ids)):
            r_mod, w_mod = get_diff_ids(r[1:].tolist(), w[1:].tolist(), min_match_size=3)
            win_token_weight[idx][w_mod] = self.mod_token_weight
            rej_token_weight[idx][r_mod] = self.mod_token_weight
        batch["win_token_weight"] = win_token_weight
        batch["rej_token_weight"] = rej_token_weight
        batch["concatenated_token_weight"] = concate_pad(win_token_weight, rej_token_weight, 0)
        for ins in win_instances + rej_instances:
            assert len(ins["input_ids"]) == len(ins["labels"])
        if torch.any(torch.isnan(batch["win_token_weight"])) or torch.any(torch.isnan(batch["rej_token_weight"])):
            raise FloatingPointError("token weight is NaN")      # reference: print + exit()
        return batch


class SyntheticPreferenceDataset(torch.utils.data.Dataset):
    """Seeded (image, chosen, rejected) triples of the shape BASELINE.md section 2 prescribes; returns
    (rej_dict, win_dict) exactly like DPODataset.__getitem__ so the reference collator surface is
    exercised.  No tokenizer / parquet / JPEG is involved (none exist offline)."""

    def __init__(self, n: int, vocab: int, text_len: int, prompt_len: int = 64, image_size: int = 336,
                 image_pos: int = 35, seed: int = 0, ragged: bool = False, omnilmm: Optional[dict] = None,
                 length_dist: Optional[str] = None):
        """``omnilmm`` = dict(tokens=(im_patch, im_start, im_end), num_query=, tower_tokens=, width=): the OmniLMM token
        convention (<im_start> <im_patch> x num_query <im_end> inside the prompt, omnilmm.py:221-257) and, as ``image``,
        precomputed tower tokens [tower_tokens, width] (the tower is frozen; rlaif-v_amd/omnilmm.py) - or, with
        ``pixels=True`` in the dict, a [3, image_size, image_size] pixel tensor for the model's own tower."""
        self.n, self.vocab, self.text_len, self.prompt_len = n, vocab, text_len, prompt_len
        self.image_size, self.image_pos, self.seed, self.ragged = image_size, min(image_pos, prompt_len - 2), seed, ragged
        self.omnilmm = omnilmm
        # "rlaifv": answer lengths shaped like the RLAIF-V preference data (tools/host_pipeline_bench.py synthesises rows of ~229
        # text tokens with a wide spread): chosen ~ log-normal(median 200, sigma 0.6), rejected = chosen x U(0.6, 1.4), both clipped
        # to [8, text_len - prompt_len].  None: ``ragged`` (uniform in the upper half) or full length.
        if length_dist not in (None, "rlaifv"):
            raise ValueError(f"length_dist must be None or 'rlaifv', got {length_dist!r}")
        self.length_dist = length_dist
        if omnilmm is not None and self.image_pos + omnilmm["num_query"] + 2 > prompt_len:
            raise ValueError("prompt_len too short for <im_start> + num_query patches + <im_end>")
        if omnilmm is not None and vocab > min(omnilmm["tokens"]):
            raise ValueError(f"vocab {vocab} reaches the image tokens {tuple(omnilmm['tokens'])}: random ids would plant stray "
                             "<im_patch>/<im_start>/<im_end> tokens; pass the text vocabulary (ids below the special tokens)")

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        prompt = torch.randint(3, self.vocab, (self.prompt_len,), generator=g)
        prompt[0] = 1
        if self.omnilmm is None:
            prompt[self.image_pos] = IMAGE_TOKEN_INDEX
            image = torch.randn(3, self.image_size, self.image_size, generator=g)
        else:
            o = self.omnilmm
            pt, st, en = o["tokens"]
            nq, a = o["num_query"], self.image_pos
            prompt[a], prompt[a + 1:a + 1 + nq], prompt[a + 1 + nq] = st, pt, en
            if o.get("pixels"):      # pixel input: the model runs its vision tower (rlaif-v_amd/eva_tower.py)
                image = torch.randn(3, self.image_size, self.image_size, generator=g)
            else:
                image = torch.randn(o["tower_tokens"], o["width"], generator=g).to(torch.bfloat16)
        out = []
        amax = self.text_len - self.prompt_len
        if self.length_dist == "rlaifv":
            base = float(torch.exp(math.log(200.0) + 0.6 * torch.randn((), generator=g)))
            ratio = float(0.6 + 0.8 * torch.rand((), generator=g))
            dist_len = {"win": int(min(max(base, 8), amax)), "rej": int(min(max(base * ratio, 8), amax))}
        for tag in ("rej", "win"):
            if self.length_dist == "rlaifv":
                alen = dist_len[tag]
            elif self.ragged:
                lo = max(2, (self.text_len - self.prompt_len) // 2)
                alen = int(torch.randint(lo, self.text_len - self.prompt_len + 1, (1,), generator=g))
            else:
                alen = self.text_len - self.prompt_len
            ans = torch.randint(3, self.vocab, (alen,), generator=g)
            ans[-1] = 2
            ids = torch.cat([prompt, ans])
            lab = ids.clone()
            lab[:self.prompt_len] = IGNORE_INDEX
            d = dict(input_ids=ids, labels=lab, image=image)
            d[f"ref_{tag}_logp"] = -100.0 - 0.01 * i
            d[f"ref_{tag}_avg_logp"] = -1.0
            d[f"ref_{tag}_per_token_logp"] = [0.0] * (ids.numel() - 1)
            out.append(d)
        return out[0], out[1]
