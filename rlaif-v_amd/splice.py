"""Host-side integer planning for the image splice and the label-driven row selection.

Mirrors ``LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal``
(/root/reference llava/model/llava_arch.py:150-330) for the DPO call site, where
``attention_mask=None`` (muffin/train/trainers.py:199, so pads are kept: llava_arch.py:220-231) and
``get_batch_logps``' shift / mask (muffin/eval/muffin_inference_logp.py:93-96).  Everything here is
int64/int32 index arithmetic on the CPU side of the batch (it replaces the reference's per-row
Python loop with its ``.tolist()`` device syncs); the tables are then consumed by the HIP kernels
``rv_splice_fwd`` / ``rv_embed_bwd`` / ``rv_feat_grad`` / ``rv_rmsnorm_*`` (row gather).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200


@dataclass
class SplicePlan:
    S: int                     # sequences (2B: wins then rejects)
    L: int                     # spliced, right-padded length
    src: torch.Tensor          # int32 [S*L]  >=0 embed row | -1 zero pad | <=-2 feature row (-2 - r)
    labels: torch.Tensor       # int64 [S, L]  new labels (IGNORE_INDEX over image span and pads)
    sel_idx: torch.Tensor      # int32 [n_sel] flat row n = s*L + l whose NEXT label is a target
    tgt: torch.Tensor          # int32 [n_sel] that target id (labels[s, l+1])
    seq_off: torch.Tensor      # int32 [S+1]   selected rows of sequence s are seq_off[s]..seq_off[s+1]
    seq_of_row: torch.Tensor   # int32 [n_sel]
    uniq_ids: torch.Tensor     # int32 [U]     distinct embedded token ids
    seg_off: torch.Tensor      # int32 [U+1]
    pos_sorted: torch.Tensor   # int32 [n_text] flat rows grouped by token id (stable order)
    feat_src_a: torch.Tensor   # int32 [n_feat_rows] flat row fed by feature row r (first user) or -1
    feat_src_b: torch.Tensor   # int32 [n_feat_rows] second user (the rejected sequence) or -1
    n_sel: int

    def to(self, device) -> "SplicePlan":
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device, non_blocking=True) if torch.is_tensor(v) else v
        return SplicePlan(**kw)


def build_splice_plan(input_ids: torch.Tensor, labels: torch.Tensor, n_img_tokens: int, n_images: int,
                      max_len: Optional[int]) -> SplicePlan:
    """input_ids/labels: int64 [S, T] (the collator's ``concatenated_*`` tensors).
    ``n_images`` distinct images were encoded (one per pair); sequence row r uses image
    ``cur_image_idx % n_images`` where cur_image_idx advances exactly like llava_arch.py:241-266
    (one per image token, and one for a row without any image token)."""
    input_ids = input_ids.cpu().long()
    labels = labels.cpu().long()
    S, T = input_ids.shape
    P = n_img_tokens
    rows_src, rows_lab = [], []
    cur_image_idx = 0
    for r in range(S):
        ids, lab = input_ids[r], labels[r]
        is_img = ids == IMAGE_TOKEN_INDEX
        n_img = int(is_img.sum())
        counts = torch.where(is_img, torch.full_like(ids, P), torch.ones_like(ids))
        tok_of = torch.repeat_interleave(torch.arange(T), counts)           # source token of each output slot
        start = torch.cumsum(counts, 0) - counts                           # first output slot of each token
        within = torch.arange(tok_of.numel()) - start[tok_of]              # offset inside an image span
        src_tok = ids[tok_of]
        out_is_img = is_img[tok_of]
        # which image (in traversal order) each image slot belongs to
        img_rank = (torch.cumsum(is_img.long(), 0) - 1)[tok_of]            # 0-based within the row
        img_global = (cur_image_idx + img_rank) % max(n_images, 1)
        src = torch.where(out_is_img, -2 - (img_global * P + within), src_tok)
        new_lab = torch.where(out_is_img, torch.full_like(src_tok, IGNORE_INDEX), lab[tok_of])
        cur_image_idx += n_img if n_img > 0 else 1
        if max_len is not None:                                            # llava_arch.py:280-283
            src, new_lab = src[:max_len], new_lab[:max_len]
        rows_src.append(src)
        rows_lab.append(new_lab)
    L = max(int(x.numel()) for x in rows_src)                              # :286
    src_full = torch.full((S, L), -1, dtype=torch.int64)                   # zero-embedding right pad (:305-313)
    lab_full = torch.full((S, L), IGNORE_INDEX, dtype=torch.int64)
    for r in range(S):
        n = rows_src[r].numel()
        src_full[r, :n] = rows_src[r]
        lab_full[r, :n] = rows_lab[r]

    # rows whose next-position label is a target (labels[:,1:] vs logits[:,:-1])
    nxt = lab_full[:, 1:]
    mask = nxt != IGNORE_INDEX                                             # [S, L-1]
    s_idx, l_idx = torch.nonzero(mask, as_tuple=True)                      # row-major order: by s then l
    sel = (s_idx * L + l_idx).to(torch.int32)
    tgt = nxt[mask].to(torch.int32)
    cnt = mask.sum(1)
    seq_off = torch.zeros(S + 1, dtype=torch.int32)
    seq_off[1:] = torch.cumsum(cnt, 0).to(torch.int32)

    flat = src_full.reshape(-1)
    text_rows = torch.nonzero(flat >= 0, as_tuple=True)[0]
    text_ids = flat[text_rows]
    order = torch.sort(text_ids, stable=True).indices
    sorted_ids = text_ids[order]
    pos_sorted = text_rows[order].to(torch.int32)
    uniq, counts_u = torch.unique_consecutive(sorted_ids, return_counts=True)
    seg_off = torch.zeros(uniq.numel() + 1, dtype=torch.int32)
    seg_off[1:] = torch.cumsum(counts_u, 0).to(torch.int32)

    n_feat_rows = max(n_images, 1) * P
    feat_rows = torch.nonzero(flat <= -2, as_tuple=True)[0]
    feat_ids = (-2 - flat[feat_rows])
    a = torch.full((n_feat_rows,), -1, dtype=torch.int32)
    b = torch.full((n_feat_rows,), -1, dtype=torch.int32)
    if feat_rows.numel():
        # users of a feature row appear in increasing flat order; with one image per sample there are
        # at most two (the chosen and the rejected sequence of the pair)
        order_f = torch.sort(feat_ids, stable=True).indices
        fid_s, frow_s = feat_ids[order_f], feat_rows[order_f]
        first = torch.ones_like(fid_s, dtype=torch.bool)
        first[1:] = fid_s[1:] != fid_s[:-1]
        second = torch.zeros_like(first)
        second[1:] = (~first[1:]) & first[:-1]
        third_plus = ~(first | second)
        if bool(third_plus.any()):
            raise ValueError("an image feature row is used by more than two sequence positions; "
                             "the DPO path expects one image per (chosen, rejected) pair")
        a[fid_s[first]] = frow_s[first].to(torch.int32)
        b[fid_s[second]] = frow_s[second].to(torch.int32)

    return SplicePlan(S=S, L=L, src=flat.to(torch.int32), labels=lab_full, sel_idx=sel, tgt=tgt, seq_off=seq_off,
                      seq_of_row=s_idx.to(torch.int32), uniq_ids=uniq.to(torch.int32), seg_off=seg_off,
                      pos_sorted=pos_sorted, feat_src_a=a, feat_src_b=b, n_sel=int(sel.numel()))
