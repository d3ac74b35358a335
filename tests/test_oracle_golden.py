"""Pins oracle/dpo_oracle.py against outputs of the reference itself (tests/golden/*.pt, produced by
tests/golden/make_golden.py from /root/reference).  CPU only."""
import os

import pytest
import torch

from oracle import dpo_oracle as O

CASES = ["tiny_b2", "tiny_b3_avg_sft", "tiny_b2_trunc", "tiny_b2_gqa"]


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name + ".pt"), weights_only=False)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_forward(golden_dir, name):
    g = _load(golden_dir, name)
    cfg = O.LlavaCfg(**g["cfg"])
    W = O.make_weights(cfg, seed=g["seed"])
    batch = O.make_synthetic_batch(cfg, g["n_pairs"], g["text_len"], g["prompt_len"], seed=g["seed"])
    with torch.no_grad():
        out = O.dpo_step_forward(batch, W, cfg, dpo_use_average=g["dpo_use_average"],
                                 sft_weight=g["sft_weight"], dpo_weight=1.0)
    # token indexing: bit exact
    assert torch.equal(out["labels"], g["labels"])
    torch.testing.assert_close(out["embeds"].double().sum(-1).float(), g["embeds_sum"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out["embeds"][0, :4], g["image_features_row0"], rtol=1e-5, atol=1e-6)
    mask = g["labels"][:, 1:] != -100
    torch.testing.assert_close(out["per_token_logps"][mask], g["per_token_logps"][mask], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out["log_prob"], g["log_prob"], rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(out["average_log_prob"], g["average_log_prob"], rtol=1e-5, atol=1e-5,
                               equal_nan=True)
    torch.testing.assert_close(out["losses"], g["losses"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out["chosen_rewards"], g["chosen_rewards"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out["rejected_rewards"], g["rejected_rewards"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out["loss"], g["loss"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", [CASES[0], CASES[1], CASES[3]])
def test_oracle_matches_reference_backward(golden_dir, name):
    g = _load(golden_dir, name)
    cfg = O.LlavaCfg(**g["cfg"])
    W = O.make_weights(cfg, seed=g["seed"])
    batch = O.make_synthetic_batch(cfg, g["n_pairs"], g["text_len"], g["prompt_len"], seed=g["seed"])
    names = O.trainable_names(cfg)
    for k in names:
        W[k].requires_grad_(True)
    out = O.dpo_step_forward(batch, W, cfg, dpo_use_average=g["dpo_use_average"],
                             sft_weight=g["sft_weight"], dpo_weight=1.0)
    out["loss"].backward()
    assert g["clip_has_grad"] is False          # CLIP tower never receives gradients (clip_encoder.py:46)
    for k, ref in g["grad_norms"].items():
        got = float(W[k].grad.double().norm())
        assert abs(got - ref) <= 2e-4 * max(ref, 1e-6) + 1e-7, (k, got, ref)
    for k, ref in g["grad_full"].items():
        torch.testing.assert_close(W[k].grad, ref, rtol=2e-3, atol=1e-6)
    torch.testing.assert_close(W["model.embed_tokens.weight"].grad.double().sum(-1).float(),
                               g["grad_embed_rowsum"], rtol=2e-3, atol=1e-6)


def test_get_batch_logps_edge_cases():
    # row without any target: log_prob 0, average NaN (muffin_inference_logp.py:104)
    logits = torch.randn(2, 5, 11)
    labels = torch.tensor([[-100, 3, 4, -100, -100], [-100] * 5])
    lp, avg = O.get_batch_logps(logits, labels)
    assert lp[1] == 0 and torch.isnan(avg[1])
    ref = logits[0].log_softmax(-1)
    assert torch.allclose(lp[0], ref[0, 3] + ref[1, 4])


def test_splice_plan_edges():
    ids = torch.tensor([[1, -200, 5, 6, 0, 0], [1, 7, -200, 8, 9, 2]])
    lab = torch.tensor([[-100, -100, 5, 6, -100, -100], [-100, -100, -100, 8, 9, 2]])
    sk, si, nl = O.splice_plan(ids, lab, 3, None)
    assert sk.shape == (2, 8)
    assert sk[0].tolist() == [1, 2, 2, 2, 1, 1, 1, 1] and si[0].tolist() == [1, 0, 1, 2, 5, 6, 0, 0]
    assert nl[1].tolist() == [-100, -100, -100, -100, -100, 8, 9, 2]
    sk2, _, nl2 = O.splice_plan(ids, lab, 3, 6)
    assert sk2.shape == (2, 6) and nl2[1].tolist() == [-100] * 5 + [8]
    # no image token in a row: plain embedding
    sk3, _, _ = O.splice_plan(torch.tensor([[1, 2, 3]]), torch.tensor([[-100, 2, 3]]), 3, None)
    assert sk3.tolist() == [[1, 1, 1]]


def test_adamw_matches_torch():
    torch.manual_seed(0)
    p = {"a.weight": torch.randn(7, 5), "b.norm.weight": torch.randn(5), "c.bias": torch.randn(3)}
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in p.items()}
    opt = torch.optim.AdamW([
        {"params": [ref["a.weight"]], "weight_decay": 0.01},
        {"params": [ref["b.norm.weight"], ref["c.bias"]], "weight_decay": 0.0}],
        lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    state = {}
    for step in range(1, 4):
        grads = {k: torch.randn_like(v) * 3 for k, v in p.items()}
        for k in ref:
            ref[k].grad = grads[k].clone()
        torch.nn.utils.clip_grad_norm_(list(ref.values()), 1.0)
        opt.step()
        O.adamw_reference(p, grads, state, 1e-3, step)
        for k in p:
            torch.testing.assert_close(p[k], ref[k].detach(), rtol=1e-5, atol=1e-6)


def test_logp_reductions_match_reference_functions(golden_dir):
    """get_batch_logps, get_batch_logps_minicpm and compute_weighted_logp of the oracle against outputs of the
    reference's own functions (tests/golden/make_logps_golden.py) - bit exact, NaN average for a target-less row."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mk", os.path.join(golden_dir, "make_logps_golden.py"))
    src = open(spec.origin).read()
    ns = {}
    exec(src[src.index("def inputs"):src.index('if __name__')], {"torch": torch}, ns)     # the input generator only
    logits, labels, weight = ns["inputs"]()
    g = torch.load(os.path.join(golden_dir, "logps_fns.pt"))
    pt, lp, avg = O.get_batch_logps(logits, labels, return_all=True)
    ptm, lpm, avgm = O.get_batch_logps_minicpm(logits, labels, return_all=True)
    for got, key in ((pt, "per_token"), (lp, "log_prob"), (avg, "avg"), (ptm, "per_token_minicpm"), (lpm, "log_prob_minicpm"),
                     (avgm, "avg_minicpm"), (O.compute_weighted_logp(pt, labels, weight, False), "weighted_sum"),
                     (O.compute_weighted_logp(pt, labels, weight, True), "weighted_avg")):
        torch.testing.assert_close(got, g[key], rtol=0, atol=0, equal_nan=True)
    assert torch.isnan(avg[3]) and torch.isnan(avgm[3])


def test_oracle_matches_reference_at_full_width(golden_dir):
    """The oracle pinned at PRODUCTION widths (d 4096, f 11008, V 32000, 32 heads, CLIP-L/14-336 at full depth, 2 LM layers)
    against the reference classes themselves (tests/golden/make_golden.py --full-width): forward and backward in one pass."""
    g = _load(golden_dir, "fullwidth_l2_b2")
    cfg = O.LlavaCfg(**g["cfg"])
    assert (cfg.hidden, cfg.ffn, cfg.vocab, cfg.clip_hidden, cfg.clip_layers, cfg.image_size) == (4096, 11008, 32000, 1024, 24, 336)
    W = O.make_weights(cfg, seed=g["seed"])
    batch = O.make_synthetic_batch(cfg, g["n_pairs"], g["text_len"], g["prompt_len"], seed=g["seed"])
    for k in O.trainable_names(cfg):
        W[k].requires_grad_(True)
    out = O.dpo_step_forward(batch, W, cfg, dpo_use_average=g["dpo_use_average"], sft_weight=g["sft_weight"], dpo_weight=1.0)
    assert torch.equal(out["labels"], g["labels"])
    mask = g["labels"][:, 1:] != -100
    torch.testing.assert_close(out["per_token_logps"][mask].detach(), g["per_token_logps"][mask], rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(out["log_prob"].detach(), g["log_prob"], rtol=2e-5, atol=0)
    torch.testing.assert_close(out["loss"].detach(), g["loss"], rtol=1e-4, atol=0)
    out["loss"].backward()
    for k, ref in g["grad_norms"].items():
        got = float(W[k].grad.double().norm())
        assert abs(got - ref) <= 5e-4 * max(ref, 1e-6) + 1e-7, (k, got, ref)
    for k, ref in g["grad_full"].items():
        torch.testing.assert_close(W[k].grad, ref, rtol=5e-3, atol=1e-6)


def test_full_depth_fixtures_carry_their_pinned_yardsticks(golden_dir):
    """The per-tensor gradient bars of the full-depth GPU tests are calibrated against the bf16-EMULATED oracle's backward, stored
    in the fixtures themselves (tests/full_depth.py compare()).  ADVICE r5: pin the committed yardsticks, so that regenerating a
    fixture - with a regressed emulated backward - cannot lower a bar unnoticed; and keep every one above the absolute floor."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import full_depth as FD
    seen = 0
    for case, worst in FD.EMU_GRAD_COS_WORST.items():
        path = os.path.join(golden_dir, f"fulldepth_{case}.pt")
        fx = torch.load(path, weights_only=False)
        got = min(fx["emu_grad_cos"].values())
        assert abs(got - worst) <= 2e-5, (case, got, worst)
        assert got >= FD.GRAD_COS_FLOOR
        seen += 1
    assert seen == 10
