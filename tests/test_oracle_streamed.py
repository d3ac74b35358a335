"""The layer-streamed evaluation of the oracle (oracle/streamed.py) equals the oracle's own one-graph step
(dpo_oracle.dpo_train_step): same forward, same gradients, also with several reference-log-prob variants off one forward.
CPU only."""
import torch

from oracle import dpo_oracle as O
from oracle import streamed as S


def _run_full(batch, W, cfg):
    W = {k: v.clone() for k, v in W.items()}
    out, grads, gn = O.dpo_train_step(batch, W, cfg, {}, lr=0.0, step=1, sft_weight=0.0, dpo_weight=1.0)
    return out, grads, gn


def test_streamed_equals_one_graph_step():
    cfg = O.tiny_cfg()
    cfg.layers = 3
    W = O.make_weights(cfg, seed=5)
    batch = O.make_synthetic_batch(cfg, 3, 48, prompt_len=12, seed=9, image_pos=5)
    out, grads, _ = _run_full(batch, W, cfg)
    got = {}
    res = S.dpo_step_streamed(batch, W, cfg, grad_sink=lambda v, n, g: got.__setitem__(n, g.clone()))
    assert torch.equal(res["labels"], out["labels"])
    assert torch.allclose(res["log_prob"], out["log_prob"].detach(), rtol=1e-6, atol=1e-5)
    assert abs(float(res["loss"]) - float(out["loss"])) <= 1e-6 * abs(float(out["loss"]))
    assert set(got) == set(grads)
    for k, g in grads.items():
        assert torch.allclose(got[k], g, rtol=1e-4, atol=1e-6 * float(g.abs().max()) + 1e-12), k


def test_streamed_variants_share_one_forward():
    cfg = O.tiny_cfg()
    W = O.make_weights(cfg, seed=6)
    batch = O.make_synthetic_batch(cfg, 2, 40, prompt_len=12, seed=3, image_pos=5)
    fwd = S.dpo_step_streamed(batch, W, cfg, backward=False)
    pw, pr = fwd["policy_win_logp"], fwd["policy_rej_logp"]
    variants = [dict(ref_win_logp=batch["ref_win_logp"], ref_rej_logp=batch["ref_rej_logp"]),
                dict(ref_win_logp=pw.clone(), ref_rej_logp=pr - torch.tensor([0.0, 5.0]))]      # beta z = 0, 0.5
    got = [{}, {}]
    res = S.dpo_step_streamed(batch, W, cfg, variants=variants, grad_sink=lambda v, n, g: got[v].__setitem__(n, g.clone()))
    assert abs(float(res["variants"][1]["losses"][0]) - 0.6931472) < 1e-5
    for v, var in enumerate(variants):
        b = dict(batch)
        b.update(var)
        _, grads, _ = _run_full(b, W, cfg)
        for k, g in grads.items():
            assert torch.allclose(got[v][k], g, rtol=1e-4, atol=1e-6 * float(g.abs().max()) + 1e-12), (v, k)


def test_streamed_row_chunks_equal_one_graph_step():
    """rows evaluated one at a time (what the L = 4096 cases need to fit the container) == the one-graph step"""
    cfg = O.tiny_cfg()
    cfg.layers = 2
    W = O.make_weights(cfg, seed=15)
    batch = O.make_synthetic_batch(cfg, 2, 44, prompt_len=10, seed=19, image_pos=5)
    out, grads, _ = _run_full(batch, W, cfg)
    got = {}
    res = S.dpo_step_streamed(batch, W, cfg, grad_sink=lambda v, n, g: got.__setitem__(n, g.clone()), row_chunk=1)
    assert torch.allclose(res["log_prob"], out["log_prob"].detach(), rtol=1e-6, atol=1e-5)
    assert torch.allclose(res["per_token_logps"], out["per_token_logps"].detach(), rtol=1e-5, atol=1e-5)
    assert set(got) == set(grads)
    for k, g in grads.items():
        assert torch.allclose(got[k], g, rtol=1e-4, atol=1e-6 * float(g.abs().max()) + 1e-12), k


def test_streamed_lora_with_replayed_masks_equals_one_graph_step():
    """the adapter model (base frozen, adapters + projector trainable) with per-layer dropout masks handed in layer by layer"""
    cfg = O.tiny_cfg()
    cfg.layers = 3
    W = O.make_weights(cfg, seed=7)
    W.update(O.make_lora_weights(cfg, 8, seed=3))
    batch = O.make_synthetic_batch(cfg, 2, 40, prompt_len=12, seed=4, image_pos=5)
    Sx, L = 4, batch["concatenated_input_ids"].shape[1] - 1 + cfg.n_patches
    g = torch.Generator().manual_seed(0)
    dims = {"self_attn.q_proj": cfg.hidden, "self_attn.k_proj": cfg.hidden, "self_attn.v_proj": cfg.hidden, "self_attn.o_proj": cfg.hidden,
            "mlp.gate_proj": cfg.hidden, "mlp.up_proj": cfg.hidden, "mlp.down_proj": cfg.ffn}
    masks = {f"model.layers.{i}.{t}": (torch.rand(Sx, L, w, generator=g) >= 0.25).float() / 0.75
             for i in range(cfg.layers) for t, w in dims.items()}
    Wc = {k: v.clone() for k, v in W.items()}
    out, grads, _ = O.dpo_train_step(batch, Wc, cfg, {}, lr=0.0, step=1, sft_weight=0.0, dpo_weight=1.0, lora_scale=0.5, lora_masks=masks)
    got = {}
    res = S.dpo_step_streamed(batch, W, cfg, grad_sink=lambda v, n, g: got.__setitem__(n, g.clone()), lora_scale=0.5, row_chunk=2,
                              lora_masks_fn=lambda i: {k: v for k, v in masks.items() if k.startswith(f"model.layers.{i}.")})
    assert torch.allclose(res["log_prob"], out["log_prob"].detach(), rtol=1e-6, atol=1e-5)
    assert set(got) == set(grads) == set(O.lora_trainable_names(W))
    for k, gr in grads.items():
        assert torch.allclose(got[k], gr, rtol=1e-4, atol=1e-6 * float(gr.abs().max()) + 1e-12), k


def test_streamed_omnilmm_front_equals_one_graph():
    """OmniLMM front (Resampler + replacement splice, grouped-query decoder) == omnilmm_step_forward + autograd"""
    from oracle import omnilmm_oracle as OO
    cfg = O.tiny_gqa_cfg()
    cfg.layers = 2
    nq, kvd, heads_rs = 4, 24, 2
    tokens = (cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1)
    W = {k: v for k, v in O.make_weights(cfg, seed=21).items() if "vision_tower" not in k and "mm_projector" not in k}
    W.update(OO.make_resampler_weights(cfg.hidden, kvd, nq, seed=2))
    batch = OO.make_omnilmm_batch(cfg, 2, 36, nq, tokens, seed=8)
    tok = torch.randn(2, 9, kvd, generator=torch.Generator().manual_seed(1))
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    ref = OO.omnilmm_step_forward(batch, tok, Wg, cfg, heads_rs, tokens)
    ref["loss"].backward()
    got = {}
    front = S.OmniLMMFront(batch, tok, W, heads_rs, tokens)
    res = S.dpo_step_streamed(batch, W, cfg, grad_sink=lambda v, n, g: got.__setitem__(n, g.clone()), front=front, row_chunk=3)
    assert torch.allclose(res["log_prob"], ref["log_prob"].detach(), rtol=1e-6, atol=1e-5)
    assert abs(float(res["loss"]) - float(ref["loss"])) <= 1e-6 * abs(float(ref["loss"])) + 1e-7
    assert set(got) == {k for k, v in Wg.items() if v.grad is not None}
    for k in got:
        gr = Wg[k].grad
        assert torch.allclose(got[k], gr, rtol=1e-4, atol=1e-6 * float(gr.abs().max()) + 1e-12), k


def test_rounding_study_layer_is_the_oracle_layer_when_no_point_is_selected():
    """oracle/rounding.py (the rounding-point attribution of round 5): with no point selected its decoder layer, its CLIP tower and its
    front ARE the oracle's (bit for bit), with every point selected the log-probs move by a bf16-sized amount, and the streamed runner
    accepts the hooks (layer_fn / hidden_fn / front) forward-only."""
    from oracle import rounding as RD
    cfg = O.tiny_cfg()
    cfg.layers = 2
    W = O.make_weights(cfg, seed=31)
    batch = O.make_synthetic_batch(cfg, 2, 40, prompt_len=12, seed=5, image_pos=5)
    base = S.dpo_step_streamed(batch, W, cfg, backward=False)
    none = S.dpo_step_streamed(batch, W, cfg, backward=False, layer_fn=RD.make_layer_fn(""), hidden_fn=RD.make_hidden_fn(""),
                               front=RD.RoundedLlavaFront(batch, cfg, W, ""))
    assert torch.equal(none["per_token_logps"], base["per_token_logps"])
    res_none = S.dpo_step_streamed(batch, W, cfg, backward=False, layer_fn=RD.make_layer_fn(""), front=RD.ResolvedLlavaFront(batch, cfg, W, ""))
    assert torch.allclose(res_none["per_token_logps"], base["per_token_logps"], rtol=0, atol=1e-5)
    px = torch.cat([batch["images"], batch["images"]], 0)
    assert torch.equal(RD.clip_features_rounded(px, W, cfg, ""), O.clip_vision_features(px, W, cfg))
    allp = S.dpo_step_streamed(batch, W, cfg, backward=False, layer_fn=RD.make_layer_fn(RD.ALL_POINTS),
                               hidden_fn=RD.make_hidden_fn(RD.ALL_POINTS), front=RD.RoundedLlavaFront(batch, cfg, W, RD.ALL_POINTS))
    mask = base["labels"][:, 1:] != O.IGNORE_INDEX
    d = (allp["per_token_logps"] - base["per_token_logps"])[mask].abs()
    assert 1e-5 < float(d.mean()) < 5e-2
    only_r = S.dpo_step_streamed(batch, W, cfg, backward=False, layer_fn=RD.make_layer_fn("R"))
    d_r = (only_r["per_token_logps"] - base["per_token_logps"])[mask].abs()
    assert 0 < float(d_r.mean()) < float(d.mean())


def test_multistep_mixed_precision_oracle_equals_one_graph_steps(tmp_path):
    """tests/full_depth.py ``oracle_multistep`` (round 6: T optimisation steps in the reference's --bf16 + ZeRO-2 arrangement - fp32
    arithmetic on bf16(master), AdamW on fp32 masters held on disk-backed arrays, the streamed evaluation reading the parameters
    THROUGH a rounding view) == the same three steps done the plain way: one autograd graph per step on explicitly rounded copies
    and dpo_oracle.adamw_reference over the whole state.  Also pins the outlier-channel construction's contract at tiny size."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import full_depth as FD
    cfg = O.tiny_cfg()
    FD.CASES["_tiny_base"] = dict(seed=1, pairs=2, text_len=40, prompt_len=12, ragged=False, answer_lens=[(20, 10), (15, 8)], lr=1e-3, step=True)
    FD.CASES["_tiny_ms"] = dict(base="_tiny_base", multistep=True, seeds=[1, 2, 3], lr=1e-3, step=True)
    try:
        W = O.make_weights(cfg, seed=3)
        W2 = {k: v.clone() for k, v in W.items()}
        fx = FD.oracle_multistep("_tiny_ms", W, cfg, str(tmp_path), log=lambda *a: None)
        assert not os.listdir(tmp_path)                         # the 3 x 27 GB scratch arrays of a real run are removed
        state = {}
        for t, batch in enumerate(FD.make_step_batches("_tiny_ms", cfg), start=1):
            Wc = {k: v.to(torch.bfloat16).float() for k, v in W2.items()}
            leaves = {k: Wc[k].clone().requires_grad_(True) for k in O.trainable_names(cfg)}
            out = O.dpo_step_forward(batch, dict(Wc, **leaves), cfg, sft_weight=0.0, dpo_weight=1.0)
            out["loss"].backward()
            grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
            tot = O.adamw_reference({k: W2[k] for k in grads}, grads, state, 1e-3, t)
            s = fx["steps"][t - 1]
            assert abs(float(out["loss"]) - s["loss"]) <= 1e-6 * abs(s["loss"]) and abs(tot - s["grad_norm_total"]) <= 1e-5 * tot
            for k in s["post_samples"]:
                idx = FD.sample_index(k, W2[k].numel())
                assert torch.allclose(W2[k].flatten()[idx], s["post_samples"][k], rtol=0, atol=1e-7), (t, k)
                assert torch.allclose(state[k]["m"].flatten()[idx], s["m_samples"][k], rtol=1e-4, atol=1e-12), (t, k)
                assert torch.allclose(state[k]["v"].flatten()[idx], s["v_samples"][k], rtol=1e-4, atol=1e-16), (t, k)
        # the masters moved off the bf16 grid (otherwise the rounding view would be vacuous)
        k = "model.layers.0.mlp.down_proj.weight"
        assert not torch.equal(W[k], W[k].to(torch.bfloat16).float())
    finally:
        del FD.CASES["_tiny_base"], FD.CASES["_tiny_ms"]
