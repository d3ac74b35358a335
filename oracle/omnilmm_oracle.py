"""CPU restatement of the OmniLMM vision-to-language path (BASELINE config 4).

TEST INFRASTRUCTURE ONLY: imported by tests/, tests/golden/make_omnilmm_golden.py and nothing else; the product path
(rlaif-v_amd/) never imports it.

What is restated, with the reference lines it follows:
  * ``get_2d_sincos_pos_embed`` / ``get_abs_pos``            omnilmm/model/resampler.py:22-93
  * ``Resampler.forward``                                     omnilmm/model/resampler.py:96-168 (kv_proj -> ln_kv, ln_q(query),
    ``nn.MultiheadAttention`` with q = ln_q(query) + pos_embed, k = x + interpolated pos_embed, v = x, ln_post, ``@ proj``)
  * the ``<im_start> <im_patch> x num_query <im_end>`` replacement splice   omnilmm/model/omnilmm.py:221-257
  * ``forward_DPO`` with ``get_batch_logps`` on the UNCHANGED labels        muffin/train/trainers.py:66-88
The language model underneath is the Mistral decoder = the Llama arithmetic with grouped-query attention, i.e.
``oracle.dpo_oracle.llama_logits`` (pinned by tests/golden/tiny_b2_gqa.pt).

Pinned: tests/golden/omnilmm_tiny.pt is produced by the reference's own ``OmniLMMForCausalLM`` / ``Resampler`` classes
(tests/golden/make_omnilmm_golden.py; timm and torchvision are absent offline, so the EVA02 tower is replaced by a stub
feature extractor there - everything FROM the tower features on is the reference's code).  The EVA02-E/14 tower itself
(timm ``eva02_enormous_patch14_clip_224``, not vendored) is therefore "parity unpinned".
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import dpo_oracle as O

RS = "model.resampler."


def get_1d_sincos_pos_embed_from_grid(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    omega = np.arange(embed_dim // 2, dtype=np.float32)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim: int, grid_size: int) -> np.ndarray:
    """resampler.py:42-73: [grid_size**2, embed_dim]; first half of the channels encodes grid[0] (= the w index), the
    second half grid[1]."""
    grid_h = np.arange(grid_size, dtype=np.float32)
    grid_w = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, grid_size, grid_size])
    emb_h = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0])
    emb_w = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)


def get_abs_pos(abs_pos: torch.Tensor, tgt_size: int) -> torch.Tensor:
    """resampler.py:22-39: bicubic resize of the [G*G, C] table to sqrt(tgt_size)**2 rows when the sizes differ."""
    src = int(math.sqrt(abs_pos.size(0)))
    tgt = int(math.sqrt(tgt_size))
    if src == tgt:
        return abs_pos
    return F.interpolate(abs_pos.float().reshape(1, src, src, -1).permute(0, 3, 1, 2), size=(tgt, tgt), mode="bicubic",
                         align_corners=False).permute(0, 2, 3, 1).flatten(0, 2).to(abs_pos.dtype)


def resampler_weight_shapes(d: int, kv_dim: int, num_query: int) -> Dict[str, Tuple[int, ...]]:
    s = {RS + "query": (num_query, d), RS + "kv_proj.weight": (d, kv_dim), RS + "attn.in_proj_weight": (3 * d, d),
         RS + "attn.in_proj_bias": (3 * d,), RS + "attn.out_proj.weight": (d, d), RS + "attn.out_proj.bias": (d,),
         RS + "proj": (d, d)}
    for n in ("ln_q", "ln_kv", "ln_post"):
        s[RS + n + ".weight"] = (d,)
        s[RS + n + ".bias"] = (d,)
    return s


def make_resampler_weights(d: int, kv_dim: int, num_query: int, seed: int = 7, bf16_round: bool = True) -> Dict[str, torch.Tensor]:
    out = {}
    for idx, (k, shp) in enumerate(resampler_weight_shapes(d, kv_dim, num_query).items()):
        g = torch.Generator().manual_seed(seed * 1000003 + idx)
        if ".ln_" in k and k.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            t = 0.02 * torch.randn(shp, generator=g)
        elif k.endswith("proj") and not k.endswith("kv_proj"):
            t = (d ** -0.5) * torch.randn(shp, generator=g)          # resampler.py:131-132
        else:
            t = 0.02 * torch.randn(shp, generator=g)
        out[k] = t.to(torch.bfloat16).to(torch.float32) if bf16_round else t
    return out


def resampler_forward(x: torch.Tensor, W: Dict[str, torch.Tensor], num_heads: int, eps: float = 1e-6) -> torch.Tensor:
    """x: [B, N, kv_dim] tower tokens (prefix tokens already stripped, omnilmm.py:113-118) -> [B, num_query, d]."""
    B, N, _ = x.shape
    query = W[RS + "query"]
    nq, d = query.shape
    hd = d // num_heads
    pos_q = torch.from_numpy(get_2d_sincos_pos_embed(d, int(math.sqrt(nq)))).float()
    pos_k = get_abs_pos(pos_q, N).to(query.dtype)           # (dtype-generic: the bf16-emulated run keeps the tables in the model dtype,
    pos_q = pos_q.to(query.dtype)                           #  resampler.py:143-149 adds them to bf16 activations under --bf16)
    xp = F.linear(x, W[RS + "kv_proj.weight"])
    xk = F.layer_norm(xp, (d,), W[RS + "ln_kv.weight"], W[RS + "ln_kv.bias"], eps)
    qn = F.layer_norm(query, (d,), W[RS + "ln_q.weight"], W[RS + "ln_q.bias"], eps)
    wi, bi = W[RS + "attn.in_proj_weight"], W[RS + "attn.in_proj_bias"]
    q = F.linear(qn + pos_q, wi[:d], bi[:d])                          # [nq, d], the same for every image
    k = F.linear(xk + pos_k, wi[d:2 * d], bi[d:2 * d])               # [B, N, d]
    v = F.linear(xk, wi[2 * d:], bi[2 * d:])
    qh = q.view(nq, num_heads, hd).transpose(0, 1)[None]             # [1, H, nq, hd]
    kh = k.view(B, N, num_heads, hd).transpose(1, 2)
    vh = v.view(B, N, num_heads, hd).transpose(1, 2)
    att = torch.softmax((qh @ kh.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = (att @ vh).transpose(1, 2).reshape(B, nq, d)
    o = F.linear(o, W[RS + "attn.out_proj.weight"], W[RS + "attn.out_proj.bias"])
    y = F.layer_norm(o, (d,), W[RS + "ln_post.weight"], W[RS + "ln_post.bias"], eps)
    return y @ W[RS + "proj"]


def omnilmm_splice(input_ids: torch.Tensor, embed_weight: torch.Tensor, image_features: torch.Tensor, im_patch: int,
                   im_start: int, im_end: int) -> torch.Tensor:
    """omnilmm.py:221-257 (use_im_start_end, orig_embeds_params None): the num_query embeddings that follow an <im_start>
    are REPLACED by the image's features; the sequence length and the labels do not change.  ``cur_image_idx`` advances
    once per <im_start> and not at all for a row without <im_patch> tokens; a row with several images keeps only the last
    replacement (each one restarts from the original row - the reference's behaviour, kept)."""
    embeds = F.embedding(input_ids, embed_weight)
    out = []
    cur = 0
    for ids, e in zip(input_ids, embeds):
        if int((ids == im_patch).sum()) == 0:
            out.append(e)
            continue
        starts = torch.where(ids == im_start)[0]
        if starts.numel() != int((ids == im_end).sum()):
            raise ValueError("The number of image start tokens and image end tokens should be the same.")
        new = e
        for p in starts.tolist():
            f = image_features[cur]
            n = f.shape[0]
            if int(ids[p + n + 1]) != im_end:
                raise ValueError("The image end token should follow the image start token.")
            new = torch.cat([e[:p + 1], f, e[p + n + 1:]], dim=0)
            cur += 1
        out.append(new)
    return torch.stack(out, dim=0)


def omnilmm_step_forward(batch: Dict[str, object], tower_features: torch.Tensor, W: Dict[str, torch.Tensor], cfg: O.LlavaCfg,
                         num_heads_resampler: int, tokens: Tuple[int, int, int], dpo_use_average: bool = False
                         ) -> Dict[str, torch.Tensor]:
    """get_beta_and_logps (trainers.py:161-275) on the generic branch: images = cat([images, images]) (:190), forward_DPO
    (:66-88) = model(input_ids, labels, images) -> get_batch_logps(logits, labels), then dpo_loss / the loss mix.
    ``tower_features``: [B, N, kv_dim] for the B distinct images.  tokens = (im_patch, im_start, im_end)."""
    feats = resampler_forward(torch.cat([tower_features, tower_features], 0), W, num_heads_resampler)
    embeds = omnilmm_splice(batch["concatenated_input_ids"], W["model.embed_tokens.weight"], feats, *tokens)
    logits = O.llama_logits(embeds, W, cfg)
    labels = batch["concatenated_labels"]
    per_token, log_prob, avg = O.get_batch_logps(logits, labels, return_all=True)
    cat = avg if dpo_use_average else log_prob
    B = batch["win_input_ids"].shape[0]
    pw, pr = cat.split([B, B])
    rw = batch["ref_win_avg_logp"] if dpo_use_average else batch["ref_win_logp"]
    rr = batch["ref_rej_avg_logp"] if dpo_use_average else batch["ref_rej_logp"]
    losses, cw, cr = O.dpo_loss(pw, pr, rw, rr, batch["beta"])
    return dict(loss=losses.mean(), losses=losses, chosen_rewards=cw, rejected_rewards=cr, policy_win_logp=pw,
                policy_rej_logp=pr, per_token_logps=per_token, log_prob=log_prob, average_log_prob=avg, embeds=embeds,
                image_features=feats, logits=logits)


def make_omnilmm_batch(cfg: O.LlavaCfg, n_pairs: int, text_len: int, num_query: int, tokens: Tuple[int, int, int],
                       prompt_len: int = 8, seed: int = 0, answer_lens=None) -> Dict[str, torch.Tensor]:
    """Synthetic preference batch in the OmniLMM token convention: [bos, prompt.., <im_start>, <im_patch> x nq, <im_end>,
    question.., answer..]; chosen and rejected share everything up to the answer; ragged answer lengths, right padded with
    id 0 / label -100 (the collator's convention, muffin/train/train_muffin.py:43-112)."""
    im_patch, im_start, im_end = tokens
    g = torch.Generator().manual_seed(seed)
    lo, hi = 3, cfg.vocab - 4
    wins, rejs, wl, rl = [], [], [], []
    for b in range(n_pairs):
        prompt = torch.randint(lo, hi, (prompt_len,), generator=g)
        img = torch.cat([torch.tensor([im_start]), torch.full((num_query,), im_patch), torch.tensor([im_end])])
        quest = torch.randint(lo, hi, (6,), generator=g)
        head = torch.cat([torch.tensor([1]), prompt, img, quest])
        n_ans = text_len - head.numel()
        la = n_ans - int(torch.randint(0, max(n_ans // 3, 1), (1,), generator=g))
        lb = n_ans - int(torch.randint(0, max(n_ans // 3, 1), (1,), generator=g))
        if answer_lens is not None:       # explicit (chosen, rejected) answer lengths (the random draws above keep the stream aligned)
            la, lb = answer_lens[b]
            assert 1 <= la <= n_ans and 1 <= lb <= n_ans
        a = torch.randint(lo, hi, (la,), generator=g)
        r = torch.randint(lo, hi, (lb,), generator=g)
        for ans, ids_l, lab_l in ((a, wins, wl), (r, rejs, rl)):
            ids = torch.cat([head, ans])
            lab = torch.cat([torch.full((head.numel(),), -100), ans])
            ids_l.append(ids)
            lab_l.append(lab)

    def pad(seqs, val, n):
        return torch.stack([torch.cat([s, torch.full((n - s.numel(),), val, dtype=s.dtype)]) for s in seqs])

    T = max(max(s.numel() for s in wins), max(s.numel() for s in rejs))
    win_ids, rej_ids = pad(wins, 0, T), pad(rejs, 0, T)
    win_lab, rej_lab = pad(wl, -100, T), pad(rl, -100, T)
    gg = torch.Generator().manual_seed(seed + 1)
    B = n_pairs
    return {"win_input_ids": win_ids, "rej_input_ids": rej_ids, "win_labels": win_lab, "rej_labels": rej_lab,
            "concatenated_input_ids": torch.cat([win_ids, rej_ids]), "concatenated_labels": torch.cat([win_lab, rej_lab]),
            "ref_win_logp": -20.0 - 5.0 * torch.rand(B, generator=gg), "ref_rej_logp": -20.0 - 5.0 * torch.rand(B, generator=gg),
            "ref_win_avg_logp": -2.0 - torch.rand(B, generator=gg), "ref_rej_avg_logp": -2.0 - torch.rand(B, generator=gg),
            "beta": 0.1}


# --------------------------------------------------------------------------------------------
# EVA02-E/14 tower: PARITY UNPINNED (timm is not vendored by the reference and absent offline).  A restatement of timm 0.9.10
# models/eva.py ``Eva(use_post_norm=True)`` as documented in rlaif-v_amd/eva_tower.py, used only to check that the HIP tower
# computes what its own docstring says.
# --------------------------------------------------------------------------------------------

def eva_weight_shapes(width: int, depth: int, heads: int, mlp: int, patch: int, pretrain_grid: int) -> Dict[str, Tuple[int, ...]]:
    s = {"patch_embed.proj.weight": (width, 3, patch, patch), "patch_embed.proj.bias": (width,), "cls_token": (1, 1, width),
         "pos_embed": (1, pretrain_grid * pretrain_grid + 1, width), "norm.weight": (width,), "norm.bias": (width,)}
    for i in range(depth):
        p = f"blocks.{i}."
        s[p + "attn.qkv.weight"] = (3 * width, width)
        s[p + "attn.q_bias"] = (width,)
        s[p + "attn.v_bias"] = (width,)
        s[p + "attn.proj.weight"] = (width, width)
        s[p + "attn.proj.bias"] = (width,)
        for n in ("norm1", "norm2"):
            s[p + n + ".weight"] = (width,)
            s[p + n + ".bias"] = (width,)
        s[p + "mlp.fc1.weight"] = (mlp, width)
        s[p + "mlp.fc1.bias"] = (mlp,)
        s[p + "mlp.fc2.weight"] = (width, mlp)
        s[p + "mlp.fc2.bias"] = (width,)
    return s


def make_eva_weights(width, depth, heads, mlp, patch, pretrain_grid, seed: int = 11) -> Dict[str, torch.Tensor]:
    out = {}
    for idx, (k, shp) in enumerate(eva_weight_shapes(width, depth, heads, mlp, patch, pretrain_grid).items()):
        g = torch.Generator().manual_seed(seed * 1000003 + idx)
        if "norm" in k and k.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            t = 0.02 * torch.randn(shp, generator=g)
        else:
            t = 0.03 * torch.randn(shp, generator=g)
        out[k] = t.to(torch.bfloat16).to(torch.float32)
    return out


def eva_forward_features(pixels: torch.Tensor, W: Dict[str, torch.Tensor], heads: int, patch: int, pretrain_grid: int,
                         blocks_used: int, eps: float = 1e-6) -> torch.Tensor:
    """[B, 3, S, S] -> [B, (S/patch)^2, width]: forward_features with the prefix token stripped (omnilmm.py:107-119)."""
    B = pixels.shape[0]
    x = F.conv2d(pixels, W["patch_embed.proj.weight"], W["patch_embed.proj.bias"], stride=patch)
    grid = x.shape[-1]
    x = x.flatten(2).transpose(1, 2)
    d = x.shape[-1]
    hd = d // heads
    pos = W["pos_embed"][0]
    if grid != pretrain_grid:
        g = pos[1:].reshape(1, pretrain_grid, pretrain_grid, d).permute(0, 3, 1, 2)
        g = F.interpolate(g, size=(grid, grid), mode="bicubic", antialias=True, align_corners=False)
        pos = torch.cat([pos[:1], g.permute(0, 2, 3, 1).reshape(grid * grid, d)], 0)
    x = torch.cat([W["cls_token"].expand(B, -1, -1), x], 1) + pos
    T = x.shape[1]
    for i in range(blocks_used):
        p = f"blocks.{i}."
        bias = torch.cat([W[p + "attn.q_bias"], torch.zeros(d), W[p + "attn.v_bias"]])
        qkv = F.linear(x, W[p + "attn.qkv.weight"], bias).view(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
        att = torch.softmax((qkv[0] @ qkv[1].transpose(-1, -2)) * hd ** -0.5, dim=-1)
        a = (att @ qkv[2]).transpose(1, 2).reshape(B, T, d)
        a = F.linear(a, W[p + "attn.proj.weight"], W[p + "attn.proj.bias"])
        x = x + F.layer_norm(a, (d,), W[p + "norm1.weight"], W[p + "norm1.bias"], eps)
        m = F.linear(F.gelu(F.linear(x, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"])), W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"])
        x = x + F.layer_norm(m, (d,), W[p + "norm2.weight"], W[p + "norm2.bias"], eps)
    x = F.layer_norm(x, (d,), W["norm.weight"], W["norm.bias"], eps)
    return x[:, 1:]
