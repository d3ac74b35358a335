"""Layer-streamed evaluation of the CPU ORACLE's DPO step.  TEST INFRASTRUCTURE ONLY (same rules as dpo_oracle.py).

``dpo_oracle.dpo_train_step`` differentiates the whole 32-layer model in one autograd graph: ~350 GB of host RAM at
LLaVA-1.5-7B widths.  This module evaluates THE SAME functions (``clip_vision_features``, ``mm_projector``,
``prepare_inputs_labels_for_multimodal``, ``llama_layer``, ``rms_norm``, ``get_batch_logps``, ``dpo_loss`` - nothing is
restated here) stage by stage with the chain rule applied by hand at the stage boundaries:

    forward   under no_grad, keeping only each layer's INPUT  [S, L, d]  (134 MB per layer at 4 x 2048 tokens);
    backward  tail (final norm, lm_head, log-probs, loss) -> d loss / d x_32; then for i = 31 .. 0 the layer is re-run
              with autograd on, differentiated against the upstream gradient, its weight gradients handed to a sink
              (norm + sampled elements) and dropped; last the front (projector + embedding + splice).

The result equals full autograd up to fp32 summation order (``tests/test_oracle_streamed.py`` compares it with
``dpo_train_step`` at small depth); peak memory is the weights (27 GB fp32) + ~10 GB, so the 7B cases run inside the
62 GB build container.  SEVERAL loss variants (different reference log-probs -> different per-row DPO coefficients)
can be differentiated off one forward: each layer is re-run once and back-propagated once per variant.

Round 5: the same runner covers the configurations that had no full-depth oracle - LoRA (adapters + projector trainable, base
frozen, the device's dropout masks replayed layer by layer: BASELINE config 5, muffin/train/train_llava15_lora.py:304-318), the
OmniLMM front (Resampler + replacement splice: BASELINE config 4, omnilmm/model/omnilmm.py:183-265) and rows evaluated in chunks
(rows of a batch are independent, so a 4 x 4096-token batch never holds more than one row's [H, L, L] attention matrices).

Reference path followed: muffin/train/trainers.py:161-311 (get_beta_and_logps + compute_loss), through the functions
of dpo_oracle.py, which cite their own lines.
"""
from __future__ import annotations

import time
from collections import ChainMap
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import dpo_oracle as O


def _overlay(W):
    """A writable view of the weight mapping: the few tensors a stage differentiates are overridden in the front map, every other
    name is read THROUGH ``W`` (its own ``__getitem__`` - the mixed-precision runs hand in a mapping that rounds the fp32 masters to
    bf16 on access, tests/full_depth.py ``ComputeView``).  For a plain dict this is what ``dict(W)`` was, without the copy."""
    return ChainMap({}, W)


def _layer_weight_names(i: int, lora: bool = False) -> List[str]:
    p = f"model.layers.{i}."
    names = [p + f"self_attn.{n}.weight" for n in ("q_proj", "k_proj", "v_proj", "o_proj")] + \
            [p + f"mlp.{n}.weight" for n in ("gate_proj", "up_proj", "down_proj")] + \
            [p + "input_layernorm.weight", p + "post_attention_layernorm.weight"]
    if lora:       # peft freezes the base model: only the adapters of the seven wrapped projections train (train_llava15_lora.py:304-318)
        names = [p + f"{t}.lora_{ab}.weight" for t in O.LORA_TARGETS for ab in ("A", "B")]
    return names


class LlavaFront:
    """CLIP tower (frozen, no_grad: clip_encoder.py:46) -> mm_projector -> embed + splice (llava_arch.py:150-330):
    what dpo_step_forward does in front of the decoder stack."""

    def __init__(self, batch, cfg: O.LlavaCfg, W, lora: bool = False):
        self.batch, self.cfg = batch, cfg
        images = batch["images"]
        with torch.no_grad():
            self.tower = O.clip_vision_features(torch.cat([images, images], dim=0), W, cfg)    # trainers.py:190 duplicates the images
        proj = [f"model.mm_projector.{j}.{k}" for j in (0, 2) for k in ("weight", "bias")]
        # LoRA run: the embedding table is frozen by peft, the projector is re-enabled (llava_arch.py:90-93)
        self.names = proj if lora else ["model.embed_tokens.weight"] + proj

    def __call__(self, W):
        feats = O.mm_projector(self.tower, W)
        x, labels = O.prepare_inputs_labels_for_multimodal(self.batch["concatenated_input_ids"], self.batch["concatenated_labels"],
                                                           feats, W["model.embed_tokens.weight"], self.cfg.model_max_length)
        return x, labels, feats


class OmniLMMFront:
    """OmniLMM (BASELINE config 4): precomputed tower tokens -> Resampler (omnilmm/model/resampler.py:96-168) -> the
    <im_start> <im_patch> x nq <im_end> REPLACEMENT splice (omnilmm/model/omnilmm.py:221-257); labels unchanged
    (forward_DPO, trainers.py:66-88).  Trainable: the embedding table and every resampler tensor (the tower is frozen in
    this path: DESIGN section 2)."""

    def __init__(self, batch, tower_features: torch.Tensor, W, num_heads_resampler: int, tokens):
        from . import omnilmm_oracle as OO
        self.OO, self.batch, self.tok, self.heads, self.tokens = OO, batch, tower_features, num_heads_resampler, tokens
        self.names = ["model.embed_tokens.weight"] + [k for k in W if k.startswith(OO.RS)]

    def __call__(self, W):
        feats = self.OO.resampler_forward(torch.cat([self.tok, self.tok], 0), W, self.heads)
        x = self.OO.omnilmm_splice(self.batch["concatenated_input_ids"], W["model.embed_tokens.weight"], feats, *self.tokens)
        return x, self.batch["concatenated_labels"], feats


def _tail_logps(x, W, labels, eps, row_chunk, hidden_fn=None):
    """per-token / sequence log-probs of ``_tail`` evaluated on row chunks (rows are independent; the [rows, L, V] fp32 logits
    of a chunk are the largest tensor of the whole run at L = 4096)."""
    S = x.shape[0]
    pts, lps, avs = [], [], []
    for r0 in range(0, S, row_chunk):
        hidden = O.rms_norm(x[r0:r0 + row_chunk], W["model.norm.weight"], eps)
        if hidden_fn is not None:
            hidden = hidden_fn(hidden)
        logits = F.linear(hidden, W["lm_head.weight"]).float()
        pt, lp, av = O.get_batch_logps(logits, labels[r0:r0 + row_chunk], return_all=True)
        pts.append(pt), lps.append(lp), avs.append(av)
        del logits, hidden
    return torch.cat(pts), torch.cat(lps), torch.cat(avs)


def dpo_step_streamed(batch: Dict[str, object], W: Dict[str, torch.Tensor], cfg: O.LlavaCfg,
                      variants=None,
                      grad_sink: Optional[Callable[[int, str, torch.Tensor], None]] = None,
                      backward: bool = True, log: Callable[[str], None] = lambda s: None,
                      timings: Optional[Dict[str, float]] = None,
                      lora_scale: Optional[float] = None,
                      lora_masks_fn: Optional[Callable[[int], Dict[str, torch.Tensor]]] = None,
                      row_chunk: Optional[int] = None, front=None,
                      layer_fn: Optional[Callable] = None, hidden_fn: Optional[Callable] = None,
                      coef_override: Optional[Sequence[torch.Tensor]] = None) -> Dict[str, object]:
    """DPO step (DPO_weight 1, SFT_weight 0, dpo_use_average False) of ``dpo_step_forward`` + ``loss.backward()``.

    variants   list of {ref_win_logp, ref_rej_logp} (or a callable (policy_win_logp, policy_rej_logp) -> such a list);
               None = the batch's own reference log-probs (one variant).
    grad_sink  called as grad_sink(variant_index, hf_name, gradient) for every trainable tensor, in backward order;
               the gradient tensor is dropped afterwards.
    lora_scale     alpha / r: the adapter model of ``dpo_step_forward(lora_scale=...)`` - base weights, embedding table, final norm and
                   lm_head frozen; adapters + projector trainable (``dpo_oracle.lora_trainable_names``).
    lora_masks_fn  layer index -> {module name: keep / (1 - p) multiplier [S, L, in]} (the dropout masks a device drew, replayed);
                   built per layer and dropped, because 32 layers of them do not fit.
    row_chunk      evaluate every stage on ``row_chunk`` batch rows at a time (rows are independent: no pad mask, no cross-row term;
                   weight gradients are summed over the chunks) - the [S, H, L, L] fp32 attention matrices of 4 rows at L = 4096
                   would not fit beside the weights otherwise.
    front          LlavaFront (default) or OmniLMMFront: everything in front of the decoder stack.
    layer_fn       replaces ``dpo_oracle.llama_layer`` (same signature) - used by the rounding-point study (oracle/rounding.py), which
                   needs the same layer with explicit bf16 roundings inserted; ``hidden_fn`` is applied to the final norm's output
                   in front of the LM head (forward-only runs).
    coef_override  per variant the vector d loss / d log_prob [S] to back-propagate instead of the one this run's own log-probs give
                   (the bf16-EMULATED backward is driven by the fp32 run's coefficients, so that the two gradients differ by the
                   backward's rounding only - the same separation tests/full_depth.py applies to the HIP path's conditioned cases).
    Returns the forward quantities of ``dpo_step_forward`` (per variant: loss / losses / rewards under ``variants``)."""
    beta = batch["beta"]
    B = batch["win_input_ids"].shape[0]
    lora = lora_scale is not None
    layer_fn = layer_fn or O.llama_layer
    t0 = time.time()
    if front is None:
        front = LlavaFront(batch, cfg, W, lora=lora)

    def run_layer(x, Wl, i, masks):
        """one decoder layer over all rows, row_chunk rows at a time"""
        S = x.shape[0]
        rc = row_chunk or S
        outs = []
        for r0 in range(0, S, rc):
            m = None if masks is None else {k: v.view(S, x.shape[1], -1)[r0:r0 + rc] for k, v in masks.items()}
            outs.append(layer_fn(x[r0:r0 + rc], Wl, cfg, i, cos, sin, causal, lora_scale, m))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    with torch.no_grad():
        x, labels, feats = front(W)
        cos, sin, causal = O.llama_tables(x.shape[1], cfg, x.dtype)
        xs = [x]
        for i in range(cfg.layers):
            masks = lora_masks_fn(i) if lora_masks_fn is not None else None
            xs.append(run_layer(xs[-1], W, i, masks))
            del masks
            if i % 8 == 7:
                log(f"forward: layer {i + 1} / {cfg.layers}, {time.time() - t0:.0f} s")
        S = xs[-1].shape[0]
        per_token, log_prob, avg = _tail_logps(xs[-1], W, labels, cfg.rms_eps, row_chunk or S, hidden_fn)
        if variants is None:
            variants = [dict(ref_win_logp=batch["ref_win_logp"], ref_rej_logp=batch["ref_rej_logp"])]
        elif callable(variants):                       # reference log-probs that depend on the policy's own (conditioned cases)
            variants = variants(log_prob[:B].float(), log_prob[B:].float())
        outs = []
        for v in variants:
            losses, cw, cr = O.dpo_loss(log_prob[:B], log_prob[B:], v["ref_win_logp"], v["ref_rej_logp"], beta)
            outs.append(dict(loss=losses.mean(), losses=losses, chosen_rewards=cw, rejected_rewards=cr))
    t1 = time.time()
    res: Dict[str, object] = dict(per_token_logps=per_token, log_prob=log_prob, average_log_prob=avg, labels=labels,
                                  policy_win_logp=log_prob[:B], policy_rej_logp=log_prob[B:], variants=outs,
                                  loss=outs[0]["loss"], image_features=feats)
    if timings is not None:
        timings["fwd_s"] = t1 - t0
    if not backward:
        return res
    if hidden_fn is not None:
        raise ValueError("hidden_fn is a forward-only hook")
    nv = len(variants)

    def sink(v, name, g):
        if grad_sink is not None and g is not None:
            grad_sink(v, name, g.detach())

    # ---- tail.  d loss / d log_prob[row] is known in closed form from dpo_loss only through autograd of the B-pair mean, so the
    # per-row coefficients are taken first (tiny graph), then each row chunk is differentiated against them.
    lp_leaf = log_prob.detach().clone().requires_grad_(True)
    coefs = []
    for v in variants:
        losses, _, _ = O.dpo_loss(lp_leaf[:B], lp_leaf[B:], v["ref_win_logp"], v["ref_rej_logp"], beta)
        coefs.append(torch.autograd.grad(losses.mean(), lp_leaf)[0])          # [S]
    if coef_override is not None:
        coefs = [c.to(log_prob.dtype) for c in coef_override]
    res["coefs"] = [c.detach().float().clone() for c in coefs]
    xL = xs.pop()
    tail_names = [] if lora else ["model.norm.weight", "lm_head.weight"]
    rc = row_chunk or S
    dxs = [torch.empty_like(xL) for _ in range(nv)]
    tail_acc = [[None] * len(tail_names) for _ in range(nv)]
    for r0 in range(0, S, rc):
        xc = xL[r0:r0 + rc].detach().requires_grad_(True)
        Wt = _overlay(W)
        leaves = []
        for n in tail_names:
            Wt[n] = W[n].detach().requires_grad_(True)
            leaves.append(Wt[n])
        hidden = O.rms_norm(xc, Wt["model.norm.weight"], cfg.rms_eps)
        logits = F.linear(hidden, Wt["lm_head.weight"]).float()
        _, lp, _ = O.get_batch_logps(logits, labels[r0:r0 + rc], return_all=True)
        for v in range(nv):
            gs = torch.autograd.grad(lp, [xc] + leaves, coefs[v][r0:r0 + rc], retain_graph=v + 1 < nv)
            dxs[v][r0:r0 + rc] = gs[0]
            for j, g in enumerate(gs[1:]):
                tail_acc[v][j] = g if tail_acc[v][j] is None else tail_acc[v][j] + g
        del logits, hidden, lp, xc, Wt, leaves, gs
    for v in range(nv):
        for n, g in zip(tail_names, tail_acc[v]):
            sink(v, n, g)
    del xL, tail_acc
    # ---- layers
    for i in reversed(range(cfg.layers)):
        names = _layer_weight_names(i, lora)
        x_all = xs.pop()
        masks = lora_masks_fn(i) if lora_masks_fn is not None else None
        acc = [[None] * len(names) for _ in range(nv)]
        for r0 in range(0, S, rc):
            x_in = x_all[r0:r0 + rc].detach().requires_grad_(True)
            Wl = _overlay(W)
            leaves = []
            for n in names:
                Wl[n] = W[n].detach().requires_grad_(True)
                leaves.append(Wl[n])
            m = None if masks is None else {k: mv.view(S, x_all.shape[1], -1)[r0:r0 + rc] for k, mv in masks.items()}
            y = layer_fn(x_in, Wl, cfg, i, cos, sin, causal, lora_scale, m)
            for v in range(nv):
                gs = torch.autograd.grad(y, [x_in] + leaves, dxs[v][r0:r0 + rc], retain_graph=v + 1 < nv)
                dxs[v][r0:r0 + rc] = gs[0]
                for j, g in enumerate(gs[1:]):
                    acc[v][j] = g if acc[v][j] is None else acc[v][j] + g
            del y, x_in, Wl, leaves, gs
        for v in range(nv):
            for n, g in zip(names, acc[v]):
                sink(v, n, g)
        del acc, masks, x_all
        if i % 8 == 0:
            log(f"backward: layer {i}, {time.time() - t1:.0f} s")
    # ---- front (the vision tower is frozen)
    fnames = list(front.names)
    Wf = _overlay(W)
    leaves = []
    for n in fnames:
        Wf[n] = W[n].detach().requires_grad_(True)
        leaves.append(Wf[n])
    emb, _, _ = front(Wf)
    for v in range(nv):
        gs = torch.autograd.grad(emb, leaves, dxs[v], retain_graph=v + 1 < nv, allow_unused=True)
        for n, g in zip(fnames, gs):
            sink(v, n, g)
    if timings is not None:
        timings["bwd_s"] = time.time() - t1
    return res
