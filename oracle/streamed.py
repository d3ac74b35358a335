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

Reference path followed: muffin/train/trainers.py:161-311 (get_beta_and_logps + compute_loss), through the functions
of dpo_oracle.py, which cite their own lines.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import dpo_oracle as O


def _layer_weight_names(i: int) -> List[str]:
    p = f"model.layers.{i}."
    return [p + f"self_attn.{n}.weight" for n in ("q_proj", "k_proj", "v_proj", "o_proj")] + \
           [p + f"mlp.{n}.weight" for n in ("gate_proj", "up_proj", "down_proj")] + \
           [p + "input_layernorm.weight", p + "post_attention_layernorm.weight"]


def _tail(x, W, labels, B, variants, beta, eps):
    """final norm + lm_head + get_batch_logps + dpo_loss for every variant; x requires grad or not."""
    hidden = O.rms_norm(x, W["model.norm.weight"], eps)
    logits = F.linear(hidden, W["lm_head.weight"]).float()
    per_token, log_prob, avg = O.get_batch_logps(logits, labels, return_all=True)
    pw, pr = log_prob.split([B, B])
    outs = []
    for v in variants:
        losses, cw, cr = O.dpo_loss(pw, pr, v["ref_win_logp"], v["ref_rej_logp"], beta)
        outs.append(dict(loss=losses.mean(), losses=losses, chosen_rewards=cw, rejected_rewards=cr))
    return per_token, log_prob, avg, outs


def dpo_step_streamed(batch: Dict[str, object], W: Dict[str, torch.Tensor], cfg: O.LlavaCfg,
                      variants=None,
                      grad_sink: Optional[Callable[[int, str, torch.Tensor], None]] = None,
                      backward: bool = True, log: Callable[[str], None] = lambda s: None,
                      timings: Optional[Dict[str, float]] = None) -> Dict[str, object]:
    """DPO step (DPO_weight 1, SFT_weight 0, dpo_use_average False) of ``dpo_step_forward`` + ``loss.backward()``.

    variants   list of {ref_win_logp, ref_rej_logp} (or a callable (policy_win_logp, policy_rej_logp) -> such a list);
               None = the batch's own reference log-probs (one variant).
    grad_sink  called as grad_sink(variant_index, hf_name, gradient) for every trainable tensor, in backward order;
               the gradient tensor is dropped afterwards.
    Returns the forward quantities of ``dpo_step_forward`` (per variant: loss / losses / rewards under ``variants``)."""
    beta = batch["beta"]
    B = batch["win_input_ids"].shape[0]
    dtype = W["model.embed_tokens.weight"].dtype
    t0 = time.time()
    with torch.no_grad():
        images = batch["images"]
        tower = O.clip_vision_features(torch.cat([images, images], dim=0), W, cfg)       # frozen, no_grad in the reference too
        feats = O.mm_projector(tower, W)
        x, labels = O.prepare_inputs_labels_for_multimodal(batch["concatenated_input_ids"], batch["concatenated_labels"], feats,
                                                           W["model.embed_tokens.weight"], cfg.model_max_length)
        cos, sin, causal = O.llama_tables(x.shape[1], cfg, x.dtype)
        xs = [x]
        for i in range(cfg.layers):
            xs.append(O.llama_layer(xs[-1], W, cfg, i, cos, sin, causal))
            if i % 8 == 7:
                log(f"forward: layer {i + 1} / {cfg.layers}, {time.time() - t0:.0f} s")
        per_token, log_prob, avg, _ = _tail(xs[-1], W, labels, B, [], beta, cfg.rms_eps)
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
    nv = len(variants)

    def sink(v, name, g):
        if grad_sink is not None and g is not None:
            grad_sink(v, name, g.detach())

    # ---- tail
    xL = xs.pop().detach().requires_grad_(True)
    tw = [W["model.norm.weight"].detach().requires_grad_(True), W["lm_head.weight"].detach().requires_grad_(True)]
    Wt = dict(W)
    Wt["model.norm.weight"], Wt["lm_head.weight"] = tw
    _, _, _, outs_g = _tail(xL, Wt, labels, B, variants, beta, cfg.rms_eps)
    dxs = []
    for v in range(nv):
        gs = torch.autograd.grad(outs_g[v]["loss"], [xL] + tw, retain_graph=v + 1 < nv)
        dxs.append(gs[0])
        sink(v, "model.norm.weight", gs[1])
        sink(v, "lm_head.weight", gs[2])
    del outs_g, xL, Wt, tw
    # ---- layers
    for i in reversed(range(cfg.layers)):
        names = _layer_weight_names(i)
        x_in = xs.pop().detach().requires_grad_(True)
        Wl = dict(W)
        leaves = []
        for n in names:
            Wl[n] = W[n].detach().requires_grad_(True)
            leaves.append(Wl[n])
        y = O.llama_layer(x_in, Wl, cfg, i, cos, sin, causal)
        for v in range(nv):
            gs = torch.autograd.grad(y, [x_in] + leaves, dxs[v], retain_graph=v + 1 < nv)
            dxs[v] = gs[0]
            for n, g in zip(names, gs[1:]):
                sink(v, n, g)
        del y, x_in, Wl, leaves, gs
        if i % 8 == 0:
            log(f"backward: layer {i}, {time.time() - t1:.0f} s")
    # ---- front: projector + embedding table + splice (the tower is frozen)
    fnames = ["model.embed_tokens.weight"] + [f"model.mm_projector.{j}.{k}" for j in (0, 2) for k in ("weight", "bias")]
    Wf = dict(W)
    leaves = []
    for n in fnames:
        Wf[n] = W[n].detach().requires_grad_(True)
        leaves.append(Wf[n])
    feats_g = O.mm_projector(tower, Wf)
    emb, _ = O.prepare_inputs_labels_for_multimodal(batch["concatenated_input_ids"], batch["concatenated_labels"], feats_g,
                                                    Wf["model.embed_tokens.weight"], cfg.model_max_length)
    for v in range(nv):
        gs = torch.autograd.grad(emb, leaves, dxs[v], retain_graph=v + 1 < nv)
        for n, g in zip(fnames, gs):
            sink(v, n, g)
    if timings is not None:
        timings["bwd_s"] = time.time() - t1
    return res
