"""Host-side integer planning for the image splice and the label-driven row selection.

Mirrors ``LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal``
(/root/reference llava/model/llava_arch.py:150-330) for the DPO call site, where
``attention_mask=None`` (muffin/train/trainers.py:199, so pads are kept: llava_arch.py:220-231) and
``get_batch_logps``' shift / mask (muffin/eval/muffin_inference_logp.py:93-96).  Everything here is
int64/int32 index arithmetic on the CPU side of the batch (it replaces the reference's per-row
Python loop with its ``.tolist()`` device syncs); the tables are then consumed by the HIP kernels
``rv_splice_fwd`` / ``rv_embed_bwd`` / ``rv_feat_grad`` / ``rv_rmsnorm_*`` (row gather).

Two layouts:
  * ``build_splice_plan``  - the reference layout: 2B rows (wins then rejects), right padded to L;
  * ``build_packed_plan``  - one row per PAIR, ``[shared prefix | chosen branch | rejected branch]``.
    The chosen and rejected sequences of a pair start with the same system prompt + image + question
    (muffin/data/datasets.py:61-63), a causal model gives those positions identical hidden states in both
    rows, so they are computed once: the rejected branch attends the shared prefix but not the chosen branch
    (rv_attn_* ``seg_sh/seg_e1``) and keeps its own RoPE positions (``pos``).  Pads are dropped (they sit to
    the right of every counted token and never influence a log-prob).  The per-sequence outputs are
    mathematically identical to the reference layout.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200


@dataclass
class SplicePlan:
    S: int                     # physical rows of the activation matrix
    L: int                     # row length (right padded)
    n_seq: int                 # logical sequences (2B: wins then rejects)
    src: torch.Tensor          # int32 [S*L]  >=0 embed row | -1 zero pad | <=-2 feature row (-2 - r)
    labels: Optional[torch.Tensor]   # int64 [S, L] new labels in the reference layout (None when packed)
    sel_idx: torch.Tensor      # int32 [n_sel] flat row n whose NEXT label is a target, ordered by (sequence, position)
    tgt: torch.Tensor          # int32 [n_sel] that target id
    seq_off: torch.Tensor      # int32 [n_seq+1] selected rows of sequence s are seq_off[s]..seq_off[s+1]
    seq_of_row: torch.Tensor   # int32 [n_sel]
    uniq_ids: torch.Tensor     # int32 [U]     distinct embedded token ids
    seg_off: torch.Tensor      # int32 [U+1]
    pos_sorted: torch.Tensor   # int32 [n_text] flat rows grouped by token id (stable order)
    feat_src_a: torch.Tensor   # int32 [n_feat_rows] flat row fed by feature row r (first user) or -1
    feat_src_b: torch.Tensor   # int32 [n_feat_rows] second user or -1
    n_sel: int
    pos: Optional[torch.Tensor] = None        # int32 [S*L] RoPE position of every row (packed layout only)
    seg_sh: Optional[torch.Tensor] = None     # int32 [S] end of the shared prefix
    seg_e1: Optional[torch.Tensor] = None     # int32 [S] end of the chosen branch
    shared_len: Optional[List[int]] = None    # per pair (accounting)
    n_real_tokens: int = 0                    # rows that carry a real token / image feature
    row_off: Optional[torch.Tensor] = None    # int32 [S] PAD-FREE packed layout: first token row of packed row s (None: s * L)
    row_len: Optional[torch.Tensor] = None    # int32 [S] ... and its length; L is then the LONGEST row (grid / lse stride / RoPE table)
    n_tokens: int = 0                         # rows of the token-major activation buffers (S * L when rectangular)

    def to(self, device) -> "SplicePlan":
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device, non_blocking=True) if torch.is_tensor(v) else v
        return SplicePlan(**kw)

    @property
    def seg(self):
        return (self.seg_sh, self.seg_e1) if self.seg_sh is not None else None

    @property
    def rows(self):
        """(row_off, row_len) for the attention kernels, None in the rectangular layouts."""
        return (self.row_off, self.row_len) if self.row_off is not None else None


def _splice_rows(input_ids: torch.Tensor, labels: torch.Tensor, n_img_tokens: int, n_images: int,
                 max_len: Optional[int]) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """Per-row spliced sources / labels before padding (llava_arch.py:237-283).  Sequence row r uses image
    ``cur_image_idx % n_images`` where cur_image_idx advances exactly like llava_arch.py:241-266 (one per image
    token, and one for a row without any image token)."""
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
        img_rank = (torch.cumsum(is_img.long(), 0) - 1)[tok_of]            # 0-based within the row
        img_global = (cur_image_idx + img_rank) % max(n_images, 1)
        src = torch.where(out_is_img, -2 - (img_global * P + within), src_tok)
        new_lab = torch.where(out_is_img, torch.full_like(src_tok, IGNORE_INDEX), lab[tok_of])
        cur_image_idx += n_img if n_img > 0 else 1
        if max_len is not None:                                            # llava_arch.py:280-283
            src, new_lab = src[:max_len], new_lab[:max_len]
        rows_src.append(src)
        rows_lab.append(new_lab)
    return rows_src, rows_lab


def make_omnilmm_splicer(im_patch: int, im_start: int, im_end: int):
    """Row splicer for the OmniLMM convention (omnilmm/model/omnilmm.py:221-257, use_im_start_end): the ``num_query``
    <im_patch> embeddings behind an <im_start> are REPLACED by the image's resampler features - no length change, labels
    untouched (the data pipeline already masks them).  ``cur_image_idx`` advances once per <im_start> and not at all for
    a row without <im_patch> tokens; the reference's batch carries cat([images, images]) (trainers.py:190), so global
    image k is distinct image k % n_images.  The reference restarts every replacement from the ORIGINAL row, so a row
    with several images keeps only the last one - reproduced here (one image per sample is the DPO data format)."""
    def splicer(input_ids, labels, n_img_tokens, n_images, max_len):
        rows_src, rows_lab = [], []
        cur = 0
        P = n_img_tokens
        for ids, lab in zip(input_ids, labels):
            src = ids.clone()
            if int((ids == im_patch).sum()) > 0:
                starts = torch.nonzero(ids == im_start, as_tuple=True)[0]
                if starts.numel() != int((ids == im_end).sum()):
                    raise ValueError("The number of image start tokens and image end tokens should be the same.")
                for p in starts.tolist():
                    if p + P + 1 >= ids.numel() or int(ids[p + P + 1]) != im_end:
                        raise ValueError("The image end token should follow the image start token.")
                    src = ids.clone()
                    src[p + 1:p + 1 + P] = -2 - ((cur % max(n_images, 1)) * P + torch.arange(P))
                    cur += 1
            if max_len is not None:
                src, lab = src[:max_len], lab[:max_len]
            rows_src.append(src)
            rows_lab.append(lab.clone())
        return rows_src, rows_lab
    return splicer


def _tables(flat: torch.Tensor, n_feat_rows: int):
    """Embedding-backward segments and feature-gradient sources from the flat source table."""
    text_rows = torch.nonzero(flat >= 0, as_tuple=True)[0]
    text_ids = flat[text_rows]
    order = torch.sort(text_ids, stable=True).indices
    sorted_ids = text_ids[order]
    pos_sorted = text_rows[order].to(torch.int32)
    uniq, counts_u = torch.unique_consecutive(sorted_ids, return_counts=True)
    seg_off = torch.zeros(uniq.numel() + 1, dtype=torch.int32)
    seg_off[1:] = torch.cumsum(counts_u, 0).to(torch.int32)

    feat_rows = torch.nonzero(flat <= -2, as_tuple=True)[0]
    feat_ids = (-2 - flat[feat_rows])
    a = torch.full((n_feat_rows,), -1, dtype=torch.int32)
    b = torch.full((n_feat_rows,), -1, dtype=torch.int32)
    if feat_rows.numel():
        # users of a feature row appear in increasing flat order; with one image per sample there are at most
        # two (the chosen and the rejected sequence of the pair; one when the image lies in the shared prefix)
        order_f = torch.sort(feat_ids, stable=True).indices
        fid_s, frow_s = feat_ids[order_f], feat_rows[order_f]
        first = torch.ones_like(fid_s, dtype=torch.bool)
        first[1:] = fid_s[1:] != fid_s[:-1]
        second = torch.zeros_like(first)
        second[1:] = (~first[1:]) & first[:-1]
        if bool((~(first | second)).any()):
            raise ValueError("an image feature row is used by more than two sequence positions; "
                             "the DPO path expects one image per (chosen, rejected) pair")
        a[fid_s[first]] = frow_s[first].to(torch.int32)
        b[fid_s[second]] = frow_s[second].to(torch.int32)
    return uniq.to(torch.int32), seg_off, pos_sorted, a, b


def build_splice_plan(input_ids: torch.Tensor, labels: torch.Tensor, n_img_tokens: int, n_images: int,
                      max_len: Optional[int], label_shift: int = 1, splicer=None) -> SplicePlan:
    """Reference layout.  input_ids/labels: int64 [S, T] (the collator's ``concatenated_*`` tensors);
    ``n_images`` distinct images were encoded (one per pair).  label_shift = 1: get_batch_logps
    (labels[:, 1:] vs logits[:, :-1], muffin_inference_logp.py:93-94); 0: get_batch_logps_minicpm (labels already
    shifted by the data pipeline: labels[:, :-1] vs logits[:, :-1], :32-33)."""
    if label_shift not in (0, 1):
        raise ValueError("label_shift must be 0 (minicpm) or 1 (llava / omnilmm)")
    input_ids = input_ids.cpu().long()
    labels = labels.cpu().long()
    S = input_ids.shape[0]
    rows_src, rows_lab = (splicer or _splice_rows)(input_ids, labels, n_img_tokens, n_images, max_len)
    L = max(int(x.numel()) for x in rows_src)                              # :286
    src_full = torch.full((S, L), -1, dtype=torch.int64)                   # zero-embedding right pad (:305-313)
    lab_full = torch.full((S, L), IGNORE_INDEX, dtype=torch.int64)
    for r in range(S):
        n = rows_src[r].numel()
        src_full[r, :n] = rows_src[r]
        lab_full[r, :n] = rows_lab[r]

    # rows whose next-position label is a target (labels[:,1:] vs logits[:,:-1]; minicpm: labels[:,:-1])
    nxt = lab_full[:, 1:] if label_shift == 1 else lab_full[:, :-1]
    mask = nxt != IGNORE_INDEX                                             # [S, L-1]
    s_idx, l_idx = torch.nonzero(mask, as_tuple=True)                      # row-major order: by s then l
    sel = (s_idx * L + l_idx).to(torch.int32)
    tgt = nxt[mask].to(torch.int32)
    seq_off = torch.zeros(S + 1, dtype=torch.int32)
    seq_off[1:] = torch.cumsum(mask.sum(1), 0).to(torch.int32)

    flat = src_full.reshape(-1)
    uniq, seg_off, pos_sorted, a, b = _tables(flat, max(n_images, 1) * n_img_tokens)
    return SplicePlan(S=S, L=L, n_seq=S, src=flat.to(torch.int32), labels=lab_full, sel_idx=sel, tgt=tgt,
                      seq_off=seq_off, seq_of_row=s_idx.to(torch.int32), uniq_ids=uniq, seg_off=seg_off,
                      pos_sorted=pos_sorted, feat_src_a=a, feat_src_b=b, n_sel=int(sel.numel()),
                      n_real_tokens=int(sum(x.numel() for x in rows_src)), n_tokens=S * L)


def build_packed_plan(input_ids: torch.Tensor, labels: torch.Tensor, n_img_tokens: int, n_images: int,
                      max_len: Optional[int], pad_token_id: int = 0, splicer=None, pad_free: bool = False) -> SplicePlan:
    """One row per pair: [shared prefix | chosen branch | rejected branch] (module docstring).
    input_ids / labels: [2B, T], wins then rejects.
    ``pad_free``: the B packed rows are CONCATENATED (row b starts at row_off[b], no inter-row padding) instead of being
    right-padded to the longest - the reference pads every row to the batch maximum (llava/model/llava_arch.py:305-313) and
    SURVEY 8a property (i) (a sequence's log-prob does not depend on its padding or its batch mates) makes dropping those rows
    exact.  Everything token-major (GEMMs, norms, SwiGLU, LM head, splice, embedding gradient) simply sees fewer rows; the
    attention kernels take (row_off, row_len) and the RoPE position table is indexed by buffer row.  Identical index tables up
    to the row renumbering: selected rows / targets / sequence offsets keep their (sequence, position) order."""
    input_ids = input_ids.cpu().long()
    labels = labels.cpu().long()
    S2 = input_ids.shape[0]
    if S2 % 2:
        raise ValueError("packed layout needs wins followed by the same number of rejects")
    B = S2 // 2
    rows_src, rows_lab = (splicer or _splice_rows)(input_ids, labels, n_img_tokens, n_images, max_len)
    for r in range(S2):        # drop the collator's right padding (pad id with label -100 at the very end)
        pad = (rows_src[r] == pad_token_id) & (rows_lab[r] == IGNORE_INDEX)
        keep = int(torch.nonzero(~pad)[-1]) + 1 if bool((~pad).any()) else 1
        rows_src[r], rows_lab[r] = rows_src[r][:keep], rows_lab[r][:keep]

    def first_target(lab):
        nz = torch.nonzero(lab != IGNORE_INDEX)
        return int(nz[0]) if nz.numel() else int(lab.numel())

    packed_src, packed_pos, shs, e1s = [], [], [], []
    sel_rows: List[List[torch.Tensor]] = [[None] * B, [None] * B]          # [branch][pair] -> row offsets in the packed row
    sel_tgts: List[List[torch.Tensor]] = [[None] * B, [None] * B]
    for b in range(B):
        cs, cl, rs, rl = rows_src[b], rows_lab[b], rows_src[B + b], rows_lab[B + b]
        n = min(cs.numel(), rs.numel())
        eq = (cs[:n] == rs[:n]) & (cl[:n] == rl[:n])
        ne = torch.nonzero(~eq)
        lcp = int(ne[0]) if ne.numel() else n
        sh = max(0, min(lcp - 1, min(first_target(cl), first_target(rl)) - 1))
        Lc, Lr = cs.numel(), rs.numel()
        packed_src.append(torch.cat([cs, rs[sh:]]))
        packed_pos.append(torch.cat([torch.arange(Lc), torch.arange(sh, Lr)]))
        shs.append(sh)
        e1s.append(Lc)
        ci = torch.nonzero(cl[1:] != IGNORE_INDEX, as_tuple=True)[0]       # row i predicts label i+1
        ri = torch.nonzero(rl[1:] != IGNORE_INDEX, as_tuple=True)[0]
        assert (ri >= sh).all() and (ci >= sh).all() or sh == 0
        sel_rows[0][b], sel_tgts[0][b] = ci, cl[1:][ci]
        sel_rows[1][b], sel_tgts[1][b] = Lc + (ri - sh), rl[1:][ri]
    L = max(int(x.numel()) for x in packed_src)
    lens = [int(x.numel()) for x in packed_src]
    if pad_free:
        starts = [0] * B
        for b in range(1, B):
            starts[b] = starts[b - 1] + lens[b - 1]
        n_tokens = starts[-1] + lens[-1]
        flat = torch.cat(packed_src)
        pos_flat = torch.cat(packed_pos)
    else:
        starts = [b * L for b in range(B)]
        n_tokens = B * L
        src_full = torch.full((B, L), -1, dtype=torch.int64)
        pos_full = torch.zeros((B, L), dtype=torch.int64)
        for b in range(B):
            n = packed_src[b].numel()
            src_full[b, :n] = packed_src[b]
            pos_full[b, :n] = packed_pos[b]
        flat, pos_flat = src_full.reshape(-1), pos_full.reshape(-1)
    sel, tgt, seq_of, counts = [], [], [], []
    for br in range(2):
        for b in range(B):
            sel.append(starts[b] + sel_rows[br][b])
            tgt.append(sel_tgts[br][b])
            seq_of.append(torch.full((sel_rows[br][b].numel(),), br * B + b, dtype=torch.int64))
            counts.append(sel_rows[br][b].numel())
    sel = torch.cat(sel).to(torch.int32)
    tgt = torch.cat(tgt).to(torch.int32)
    seq_of = torch.cat(seq_of).to(torch.int32)
    seq_off = torch.zeros(S2 + 1, dtype=torch.int32)
    seq_off[1:] = torch.cumsum(torch.tensor(counts), 0).to(torch.int32)
    uniq, seg_off, pos_sorted, a, bb = _tables(flat, max(n_images, 1) * n_img_tokens)
    return SplicePlan(S=B, L=L, n_seq=S2, src=flat.to(torch.int32), labels=None, sel_idx=sel, tgt=tgt, seq_off=seq_off,
                      seq_of_row=seq_of, uniq_ids=uniq, seg_off=seg_off, pos_sorted=pos_sorted, feat_src_a=a,
                      feat_src_b=bb, n_sel=int(sel.numel()), pos=pos_flat.to(torch.int32),
                      seg_sh=torch.tensor(shs, dtype=torch.int32), seg_e1=torch.tensor(e1s, dtype=torch.int32),
                      shared_len=shs, n_real_tokens=int(sum(lens)), n_tokens=n_tokens,
                      row_off=torch.tensor(starts, dtype=torch.int32) if pad_free else None,
                      row_len=torch.tensor(lens, dtype=torch.int32) if pad_free else None)
