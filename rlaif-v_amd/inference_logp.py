"""Reference-model log-prob precompute (the stage that runs once before DPO training).

Mirrors /root/reference muffin/eval/muffin_inference_logp.py:
  InferenceSampler                 :55-79    contiguous rank shards
  get_multimodal_sample_logps      :213-281  (is_llava15 branch) -> forward-only reuse of the training kernels
  write_logp_to_preference_parquet :283-313  same `logps` JSON column, 5000-row parquet chunks
  inference_logp                   :315-344
Differences that do not change the stored values: any batch size is accepted (the reference is pinned to 1,
:323), logits are never materialised (fused LM-head log-prob kernel), and the lists are gathered with one
all_gather_object per list exactly like the reference.
"""
from __future__ import annotations

import copy
import itertools
import json
import os
from functools import partial
from typing import List, Sequence

import torch

from .data import preference_collator_fn


class InferenceSampler(torch.utils.data.sampler.Sampler):
    """Contiguous shard of range(size) for this rank: the first size % world ranks get one more element (the reference's sampler,
    muffin/eval/muffin_inference_logp.py:55-79)."""

    def __init__(self, size: int):
        assert size > 0
        ddp = torch.distributed.is_available() and torch.distributed.is_initialized()
        rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) if ddp else (0, 1)
        self._local_indices = self._get_local_indices(int(size), world, rank)

    @staticmethod
    def _get_local_indices(total_size, world_size, rank):
        q, r = divmod(total_size, world_size)
        begin = rank * q + min(rank, r)
        return range(begin, begin + q + (rank < r))

    def __iter__(self):
        yield from self._local_indices

    def __len__(self):
        return len(self._local_indices)


class PreferenceInferenceDataset(torch.utils.data.Dataset):
    """muffin/eval/muffin_inference_logp.py:116-165: raw preference rows -> (rej_dict, win_dict) of token ids."""

    def __init__(self, data, tokenizer, image_token_len, img_processor, use_im_start_end: bool = True):
        self.data = data
        self.mm_cfg = {"image_processor": img_processor, "is_multimodal": True, "image_token_len": image_token_len,
                       "use_im_start_end": use_im_start_end, "keep_image_tag": True}
        self.tokenizer = tokenizer

    def __getitem__(self, index):
        from .dataset import bytes_to_PIL_image, encode_multimodal_preference_sample, preprocess_v1
        sample = self.data[index]
        split = sample.get("origin_split")
        metainfo = {"origin_dataset": sample.get("origin_dataset"),
                    "origin_split": json.loads(split) if isinstance(split, str) and split[:1] in "[{\"" else split,
                    "origin_idx": sample["idx"], "image_id": sample.get("image_path")}
        formated_sample = {"image": bytes_to_PIL_image(sample["image"]["bytes"]),
                           "question": {"from": "human", "value": f"<image>\n{sample['question']}"},
                           "chosen": {"from": "gpt", "value": sample["chosen"]},
                           "rejected": {"from": "gpt", "value": sample["rejected"]},
                           "idx": sample["idx"], "metainfo": metainfo}
        return encode_multimodal_preference_sample(formated_sample, self.tokenizer, self.mm_cfg,
                                                   preprocess_func=lambda s, t: preprocess_v1(s, t, has_image=True))

    def __len__(self):
        return len(self.data)


def get_multimodal_sample_logps(model, dataloader, tokenizer=None, is_llava15: bool = True):
    """Returns (win_logp, win_avg_logp, win_per_token_logp, rej_logp, rej_avg_logp, rej_per_token_logp) as Python
    lists, one entry per sample; per-token lists have spliced_length - 1 entries like the reference's."""
    if not is_llava15:
        raise NotImplementedError("only the LLaVA-1.5 branch is implemented")
    # results stay on the DEVICE while the loop runs (a few KB per row) and come back once at the end: a .tolist() per batch and
    # branch stalls the stream 2 x 83 k times on the full RLAIF-V set
    parts = {k: ([], [], []) for k in ("win", "rej")}
    model.eval()
    for batch in dataloader:
        for key in ("win", "rej"):
            input_ids, labels = batch[f"{key}_input_ids"], batch[f"{key}_labels"]
            res = model.forward_logps(input_ids, labels, batch["images"], save_for_backward=False, all_rows=True)
            S = input_ids.shape[0]
            per_tok = res.per_token_logp.view(S, -1)
            assert per_tok.size(1) >= input_ids.size(1) - 1
            parts[key][0].append(res.seq_logp.float().clone())
            parts[key][1].append((res.seq_logp / res.seq_cnt).float())
            parts[key][2].append(per_tok.float().clone())
    out = {}
    for key in ("win", "rej"):
        logp = torch.cat(parts[key][0]).tolist() if parts[key][0] else []
        avg = torch.cat(parts[key][1]).tolist() if parts[key][1] else []
        per = [row for t in parts[key][2] for row in t.cpu().tolist()]
        out[key] = (logp, avg, per)
    w, r = out["win"], out["rej"]
    return w[0], w[1], w[2], r[0], r[1], r[2]


def write_logp_to_preference_parquet(origin_data, cache_file: str, logps: Sequence, overwrite_logps: bool = False):
    import pandas as pd
    out_data = []
    for index in range(len(logps)):
        line = origin_data[index]
        new_line = copy.deepcopy(line)
        if "logps" in new_line.keys():
            assert overwrite_logps, "Found existing logp data, pass overwrite_logps=True to force overwritting"
        else:
            assert all(k in new_line.keys() for k in ("question", "chosen", "rejected")), \
                f"Undefined data structure, expecting [Q, Win, Rej] in keys, got {new_line.keys()}"
        new_line["logps"] = json.dumps({"logps": logps[index]})
        out_data.append(new_line)
    ddp = torch.distributed.is_available() and torch.distributed.is_initialized()
    if not ddp or torch.distributed.get_rank() == 0:
        os.makedirs(cache_file, exist_ok=True)
        step = 5000
        for idx, start in enumerate(range(0, len(out_data), step)):
            temp = out_data[start:min(start + step, len(out_data))]
            pd.DataFrame(temp).to_parquet(os.path.join(cache_file, f"RLAIF-V-Dataset-withlogp_{idx:03}-{len(temp)}.parquet"))
    if ddp:
        torch.distributed.barrier()


def inference_logp(model, tokenizer, dataset, cache_file: str, batch_size: int = 1, num_workers: int = 0,
                   is_llava15: bool = True):
    """``dataset`` plays the role of the reference's PreferenceInferenceDataset: ``dataset[i]`` -> (rej_dict, win_dict)
    and ``dataset.data[i]`` -> the original row (question / chosen / rejected / image ...)."""
    pad_id = getattr(tokenizer, "pad_token_id", 0) if tokenizer is not None else 0
    loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, collate_fn=partial(preference_collator_fn, pad_token_id=pad_id),
                                         num_workers=num_workers, shuffle=False, sampler=InferenceSampler(len(dataset)))
    outputs = get_multimodal_sample_logps(model, loader, tokenizer, is_llava15=is_llava15)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        world = torch.distributed.get_world_size()
        merged = []
        for o in outputs:
            buf: List = [None] * world
            torch.distributed.all_gather_object(buf, o)
            merged.append(list(itertools.chain.from_iterable(buf)))
        outputs = merged
    logps = list(zip(*outputs))
    write_logp_to_preference_parquet(dataset.data, cache_file, logps, overwrite_logps=False)
    return logps
