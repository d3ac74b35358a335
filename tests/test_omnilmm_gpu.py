"""OmniLMM branch on the GPU against the fixture the REFERENCE'S OWN OmniLMMForCausalLM /
Resampler / forward_DPO produced (tests/golden/omnilmm_tiny.pt, tests/golden/make_omnilmm_golden.py):

  * the Resampler on the HIP kernels, forward and every parameter gradient, both position-table branches;
  * the whole DPO step from tower tokens on: replacement splice (bit exact indexing), Mistral-style GQA decoder, fused
    LM-head log-probs, DPO loss, and the gradients of the language model AND the resampler; reference and packed layouts.

Bars: sequence log-probs / loss 1e-3 relative; gradients per-tensor norm within 3 %, direction cosine >= 0.995."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dpo_oracle as O  # noqa: E402
from oracle import omnilmm_oracle as OO  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "omnilmm_tiny.pt")
BF16 = torch.bfloat16


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _cfg(meta):
    from rlaif_v_amd.omnilmm import OmniLMMConfig
    pt, st, en = meta["tokens"]
    return OmniLMMConfig(hidden=512, layers=2, heads=4, kv_heads=2, ffn=768, vocab=325, model_max_length=256,
                         num_query=meta["num_query"], vision_width=meta["kv_dim"], image_size=84,
                         im_patch_token=pt, im_start_token=st, im_end_token=en)


def _weights(meta):
    ocfg = O.LlavaCfg(hidden=512, layers=2, heads=4, kv_heads=2, ffn=768, vocab=325, model_max_length=256)
    W = {k: v for k, v in O.make_weights(ocfg, seed=3).items() if "vision_tower" not in k and "mm_projector" not in k}
    W.update(OO.make_resampler_weights(512, meta["kv_dim"], meta["num_query"]))
    return W


def _check_grads(gold, mine, what):
    worst = 0.0
    for k, g in gold.items():
        m = mine[k].float().cpu()
        if "full" in g:
            ref = g["full"]
            if float(ref.norm()) < 1e-7:
                continue
            c = _cos(m, ref)
            rel = abs(float(m.norm()) - float(ref.norm())) / float(ref.norm())
        else:
            blk = m.reshape(m.shape[0], -1)[:8, :64]
            c = _cos(blk, g["block"])
            rel = abs(float(m.double().norm()) - g["norm"]) / g["norm"]
        worst = max(worst, rel)
        assert c >= 0.995, (what, k, c)
        assert rel <= 3e-2, (what, k, rel)
    print(f"[{what}] worst per-tensor grad-norm rel err {worst:.3e}")


@pytest.mark.parametrize("n_tok", [36, 16])
def test_resampler_matches_reference_golden(n_tok):
    _need_gpu()
    from rlaif_v_amd.resampler import Resampler, param_shapes
    G = torch.load(GOLD, weights_only=False)
    meta, case = G["meta"], G[f"resampler_{n_tok}"]
    W = _weights(meta)
    dev = torch.device("cuda:0")
    rs = Resampler(512, meta["kv_dim"], meta["num_query"], dev)
    P = {k: W[OO.RS + k].to(dev, BF16).contiguous() for k in param_shapes(512, meta["kv_dim"], meta["num_query"])}
    Gd = {k: torch.zeros_like(v) for k, v in P.items()}
    B, N, kv = case["x"].shape
    ctx = {}
    z = rs.forward(case["x"].to(dev, BF16).reshape(B * N, kv).contiguous(), B, N, P.__getitem__, ctx)
    y = z.float().cpu().view(B, meta["num_query"], 512)
    err = float((y - case["y"]).abs().max()) / float(case["y"].abs().max())
    print(f"resampler n_tok={n_tok}: forward max err / max |y| = {err:.3e}")
    assert err <= 2e-2 and _cos(y, case["y"]) >= 0.9995
    rs.backward(case["gy"].to(dev, BF16).reshape(B * meta["num_query"], 512).contiguous(), ctx, P.__getitem__, Gd.__getitem__)
    _check_grads(case["grads"], Gd, f"resampler {n_tok}")


@pytest.mark.parametrize("share_prefix", [False, True])
def test_dpo_step_matches_reference_golden(share_prefix, monkeypatch):
    _need_gpu()
    from rlaif_v_amd.omnilmm import OmniLMMDPOModel
    from rlaif_v_amd.trainer import LLaVA15DPOTrainer, TrainingArguments
    monkeypatch.setenv("SFT_weight", "0.0")
    monkeypatch.setenv("DPO_weight", "1.0")
    G = torch.load(GOLD, weights_only=False)
    meta, case = G["meta"], G["dpo"]
    model = OmniLMMDPOModel(_cfg(meta))
    model.share_prefix = share_prefix
    model.load_state_dict(_weights(meta))
    tr = LLaVA15DPOTrainer(model=model, args=TrainingArguments())
    batch = dict(case["batch"])
    batch["images"] = case["tower_features"]                   # precomputed tower tokens [B, N, width]
    loss = tr.compute_loss(model, batch)
    out = model.last_out
    # ---- indexing: bit exact (labels are NOT changed by the OmniLMM splice)
    labels = case["batch"]["concatenated_labels"]
    mask = labels[:, 1:] != -100
    s_idx, l_idx = torch.nonzero(mask, as_tuple=True)
    assert torch.equal(out.plan.tgt.cpu().long(), labels[:, 1:][mask])
    assert torch.equal(out.plan.seq_of_row.cpu().long(), s_idx)
    if not share_prefix:
        assert torch.equal(out.plan.labels.cpu(), labels)
        assert torch.equal(out.plan.sel_idx.cpu().long(), s_idx * labels.shape[1] + l_idx)
        st = int(torch.where(case["batch"]["concatenated_input_ids"][0] == meta["tokens"][1])[0][0])
        src = out.plan.src.cpu().view(out.plan.S, out.plan.L)
        assert src[0, st] == meta["tokens"][1] and src[0, st + 1] == -2 and src[0, st + meta["num_query"]] == -2 - (meta["num_query"] - 1)
        assert src[2, st + 1] == -2                              # rejected row of pair 0 reads the same image
    else:
        assert out.plan.S == 2 and max(out.plan.shared_len) > meta["num_query"]       # the image lies in the shared prefix
    ref_lp = case["logp"]
    err_lp = (out.seq_logp.cpu() - ref_lp).abs()
    print(f"omnilmm share_prefix={share_prefix}: seq logp err {err_lp.tolist()} of {ref_lp.tolist()}; loss {float(loss):.6f} "
          f"vs {float(case['loss']):.6f}")
    assert bool((err_lp <= 1e-3 * ref_lp.abs()).all())
    torch.testing.assert_close(loss.cpu(), case["loss"], rtol=1e-3, atol=0.0)
    torch.testing.assert_close(out.per_pair[1].cpu(), case["chosen_rewards"], rtol=2e-3, atol=1e-2)
    # ---- backward: language model + resampler gradients
    model.backward(out, model.last_coef)
    _check_grads(case["grads"], model.grads_state_dict(), f"dpo share_prefix={share_prefix}")


def test_pixel_input_without_tower_fails_loudly():
    _need_gpu()
    from rlaif_v_amd.omnilmm import OmniLMMDPOModel
    G = torch.load(GOLD, weights_only=False)
    model = OmniLMMDPOModel(_cfg(G["meta"]), with_optimizer=False)
    with pytest.raises(NotImplementedError):
        model.encode_images(torch.zeros(1, 3, 84, 84))


def test_eva_tower_matches_its_restatement_and_feeds_the_model():
    """EVA02 tower: PARITY UNPINNED (see rlaif-v_amd/eva_tower.py) - the HIP tower against the author's own torch restatement
    (head dim 112 on the 128-wide kernels, post-norm blocks, resampled positions), then pixels -> tower -> resampler -> DPO
    forward through ``set_vision_tower`` equals the same forward fed with the tower tokens."""
    _need_gpu()
    from rlaif_v_amd.eva_tower import EvaConfig, EvaTower
    from rlaif_v_amd.omnilmm import OmniLMMDPOModel
    G = torch.load(GOLD, weights_only=False)
    meta, case = G["meta"], G["dpo"]
    ec = EvaConfig(width=448, depth=4, heads=4, mlp=768, patch=14, pretrain_grid=3)     # head dim 112, as EVA02-E
    We = OO.make_eva_weights(ec.width, ec.depth, ec.heads, ec.mlp, ec.patch, ec.pretrain_grid)
    tower = EvaTower(ec)
    tower.load_state_dict(We)
    px = torch.randn(2, 3, 84, 84, generator=torch.Generator().manual_seed(2))        # 6 x 6 grid: positions are resampled from 3 x 3
    ref = OO.eva_forward_features(px, We, ec.heads, ec.patch, ec.pretrain_grid, ec.blocks_used)
    got = tower(px).float().cpu()
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    print(f"eva tower: max err / max |ref| = {err:.3e}, cosine {_cos(got, ref):.6f}")
    assert got.shape == ref.shape == (2, 36, 448)
    assert err <= 3e-2 and _cos(got, ref) >= 0.9995
    # pixels in, through the model
    cfg = _cfg(meta)
    cfg.vision_width = 448
    model = OmniLMMDPOModel(cfg, with_optimizer=False)
    W = _weights(meta)
    W.update(OO.make_resampler_weights(512, 448, meta["num_query"]))
    model.load_state_dict(W)
    model.set_vision_tower(tower)
    b = case["batch"]
    out_px = model.forward_logps(b["concatenated_input_ids"], b["concatenated_labels"], px, save_for_backward=False)
    out_tok = model.forward_logps(b["concatenated_input_ids"], b["concatenated_labels"], tower(px), save_for_backward=False)
    assert torch.equal(out_px.seq_logp, out_tok.seq_logp)


def test_omnilmm_full_width_shallow_vs_oracle(monkeypatch):
    """OmniLMM-12B widths (Mistral-7B: d 4096, f 14336, 32 query / 8 key-value heads, the 32009-token vocabulary; Resampler
    64 queries x 1024 tower tokens of width 1792, 32 heads) with 2 decoder layers against the fp32 CPU oracle (itself pinned
    by the reference-class golden at small widths): the production tile shapes of the resampler and of the GQA decoder."""
    _need_gpu()
    from rlaif_v_amd.omnilmm import OmniLMMConfig, OmniLMMDPOModel
    from rlaif_v_amd.trainer import LLaVA15DPOTrainer, TrainingArguments
    monkeypatch.setenv("SFT_weight", "0.0")
    monkeypatch.setenv("DPO_weight", "1.0")
    cfg = OmniLMMConfig(layers=2, model_max_length=256)
    ocfg = O.LlavaCfg(hidden=cfg.hidden, layers=2, heads=cfg.heads, kv_heads=cfg.n_kv_heads, ffn=cfg.ffn, vocab=cfg.vocab,
                      model_max_length=256)
    W = {k: v for k, v in O.make_weights(ocfg, seed=9).items() if "vision_tower" not in k and "mm_projector" not in k}
    W.update(OO.make_resampler_weights(cfg.hidden, cfg.vision_width, cfg.num_query, seed=4))
    tokens = (cfg.im_patch_token, cfg.im_start_token, cfg.im_end_token)
    lo_cfg = O.LlavaCfg(hidden=cfg.hidden, layers=2, heads=cfg.heads, kv_heads=cfg.n_kv_heads, ffn=cfg.ffn, vocab=32000)
    batch = OO.make_omnilmm_batch(lo_cfg, 1, 112, cfg.num_query, tokens, seed=6)          # text ids below 32000
    tok = torch.randn(1, 1024, cfg.vision_width, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).float()
    # oracle first: the reference log-probs of the batch are then placed next to the policy's so that the DPO loss is O(1)
    # (random weights give sequence log-probs near -300; against the generator's -20 the loss would be exp(-28))
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    ref = OO.omnilmm_step_forward(batch, tok, Wg, ocfg, cfg.hidden // 128, tokens)
    batch["ref_win_logp"] = ref["policy_win_logp"].detach() - 3.0
    batch["ref_rej_logp"] = ref["policy_rej_logp"].detach() + 2.0
    losses, _, _ = O.dpo_loss(ref["policy_win_logp"], ref["policy_rej_logp"], batch["ref_win_logp"], batch["ref_rej_logp"],
                              batch["beta"])
    ref_loss = losses.mean()
    model = OmniLMMDPOModel(cfg)
    model.load_state_dict(W)
    tr = LLaVA15DPOTrainer(model=model, args=TrainingArguments())
    b = dict(batch)
    b["images"] = tok
    loss = tr.compute_loss(model, b)
    out = model.last_out
    err = (out.seq_logp.cpu() - ref["log_prob"].detach()).abs()
    print("omnilmm full-width shallow: seq logp", out.seq_logp.tolist(), "ref", ref["log_prob"].tolist(), "loss", float(loss),
          float(ref_loss))
    assert bool((err <= 1e-3 * ref["log_prob"].detach().abs()).all())
    # the loss is a function of beta * (difference of two sums of magnitude ~300): log-prob errors of 1e-4 RELATIVE (0.03 and
    # 0.05 absolute, measured) move it by ~1e-3 absolute = 2.3e-3 relative; the 1e-3 bar is asserted on the log-probs above
    assert abs(float(loss) - float(ref_loss)) <= 5e-3 * abs(float(ref_loss))
    ref_loss.backward()
    model.backward(out, model.last_coef)
    grads = model.grads_state_dict()
    for k in ("model.layers.1.mlp.gate_proj.weight", "model.layers.0.self_attn.k_proj.weight", "lm_head.weight",
              "model.resampler.kv_proj.weight", "model.resampler.attn.in_proj_weight", "model.resampler.query",
              "model.resampler.proj", "model.resampler.ln_kv.weight"):
        g, r = grads[k], Wg[k].grad
        c = _cos(g, r)
        rel = abs(float(g.double().norm()) - float(r.double().norm())) / float(r.double().norm())
        print(f"  grad {k}: cos {c:.5f} norm rel err {rel:.3e}")
        assert c >= 0.99 and rel <= 5e-2, (k, c, rel)
