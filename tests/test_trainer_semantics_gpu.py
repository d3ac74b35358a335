"""HF-Trainer semantics the shipped script can select, on the HIP path: --gradient_accumulation_steps (fp32 side accumulation,
one exchange per optimizer step), evaluate() (`*_test/*` metrics, forward only), `model.forward(inputs_embeds=).logits`
(trainers.py:221-225), the data position surviving a resume."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dpo_oracle as O  # noqa: E402


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _model(cfg, seed):
    from rlaif_v_amd.model import LlavaConfig, LlavaDPOModel
    model = LlavaDPOModel(LlavaConfig(**O.asdict(cfg)))
    W = O.make_weights(cfg, seed=seed)
    model.load_state_dict(W)
    return model, W


def _trainer(model, **kw):
    from rlaif_v_amd.trainer import LLaVA15DPOTrainer, TrainingArguments
    return LLaVA15DPOTrainer(model=model, args=TrainingArguments(learning_rate=1e-3, warmup_ratio=0.0, lr_scheduler_type="constant", **kw))


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _cat_batches(b1, b2, pad_id=0):
    """One batch of B1 + B2 pairs in the collator's layout (wins then rejects), right padded to the common length."""
    def pad_to(t, n, val):
        return torch.cat([t, torch.full((t.shape[0], n - t.shape[1]), val, dtype=t.dtype)], 1) if t.shape[1] < n else t
    T = max(b1["concatenated_input_ids"].shape[1], b2["concatenated_input_ids"].shape[1])
    out = {}
    for side in ("win", "rej"):
        out[f"{side}_input_ids"] = torch.cat([pad_to(b[f"{side}_input_ids"], T, pad_id) for b in (b1, b2)])
        out[f"{side}_labels"] = torch.cat([pad_to(b[f"{side}_labels"], T, -100) for b in (b1, b2)])
        for k in (f"ref_{side}_logp", f"ref_{side}_avg_logp"):
            out[k] = torch.cat([b1[k], b2[k]])
    out["concatenated_input_ids"] = torch.cat([out["win_input_ids"], out["rej_input_ids"]])
    out["concatenated_labels"] = torch.cat([out["win_labels"], out["rej_labels"]])
    out["images"] = torch.cat([b1["images"], b2["images"]])
    out["beta"] = b1["beta"]
    return out


def test_gradient_accumulation_equals_one_big_batch():
    """Two micro-batches of 2 pairs with --gradient_accumulation_steps 2 give the gradient (mean over the window) that ONE batch
    of the same 4 pairs gives, and exactly one optimizer step; the fp32 side buffer is only allocated when asked for."""
    _need_gpu()
    cfg = O.tiny_cfg()
    b1 = O.make_synthetic_batch(cfg, 2, 40, 12, seed=61)
    b2 = O.make_synthetic_batch(cfg, 2, 36, 12, seed=62)
    captured = {}

    def run(ga, batches):
        model, _ = _model(cfg, seed=8)
        tr = _trainer(model, gradient_accumulation_steps=ga)
        orig = tr.optimizer_step

        def spy(lr=None):
            captured[ga] = model.store.flat_g.clone()
            return orig(lr)
        tr.optimizer_step = spy
        losses = [tr.training_step(dict(b)) for b in batches]
        torch.cuda.synchronize()
        return tr, model, losses

    tr2, m2, l2 = run(2, [b1, b2])
    assert tr2.state["global_step"] == 1 and tr2._gacc is not None and tr2._micro == 0
    tr1, m1, l1 = run(1, [_cat_batches(b1, b2)])
    assert tr1.state["global_step"] == 1 and tr1._gacc is None
    g2, g1 = captured[2].float(), captured[1].float()
    rel = abs(float(g2.norm()) - float(g1.norm())) / float(g1.norm())
    c = _cos(g2, g1)
    print(f"grad accumulation vs one batch: cosine {c:.6f}, norm rel diff {rel:.2e}; window loss {float(l2[-1]):.5f} vs {float(l1[-1]):.5f}")
    assert c >= 0.999 and rel <= 1e-2
    assert abs(float(l2[-1]) - float(l1[-1])) <= 2e-3 * abs(float(l1[-1]))
    # the second window starts clean
    tr2.training_step(dict(b1))
    assert tr2._micro == 1 and tr2.state["global_step"] == 1


def test_evaluate_logs_test_metrics_and_keeps_weights():
    _need_gpu()
    cfg = O.tiny_cfg()
    model, W = _model(cfg, seed=9)
    from rlaif_v_amd.data import DataCollatorForDPODataset, SyntheticPreferenceDataset

    class Tok:
        pad_token_id = cfg.pad_token_id
    ds = SyntheticPreferenceDataset(n=6, vocab=cfg.vocab, text_len=40, prompt_len=12, image_size=cfg.image_size, seed=3)
    tr = _trainer(model, per_device_eval_batch_size=2)
    tr.eval_dataset, tr.data_collator = ds, DataCollatorForDPODataset(Tok(), beta=0.1, mod_token_weight=1.0)
    before = model.store.flat_p.clone()
    m = tr.evaluate()
    assert set(m) == {"rewards_test/chosen", "rewards_test/rejected", "rewards_test/accuracies", "rewards_test/margins",
                      "logps_test/chosen", "logps_test/rejected", "logps_test/ref_chosen", "logps_test/ref_rejected", "eval_loss"}
    assert all(v == v for v in m.values())                       # finite
    assert torch.equal(model.store.flat_p, before) and model.training          # nothing trained, mode restored
    assert tr.state["log_history"][-1]["eval_loss"] == m["eval_loss"]
    # mean over the three batches = what the oracle gives for the same rows
    ref_losses = []
    for i in range(0, 6, 2):
        batch = tr.data_collator([ds[i], ds[i + 1]])
        with torch.no_grad():
            ref_losses.append(float(O.dpo_step_forward(batch, W, cfg, sft_weight=0.0, dpo_weight=1.0)["loss"]))
    ref = sum(ref_losses) / 3
    assert abs(m["eval_loss"] - ref) <= 2e-3 * abs(ref), (m["eval_loss"], ref)


def test_forward_inputs_embeds_logits_match_oracle():
    """The reference's own call pair: prepare_inputs_labels_for_multimodal -> model(inputs_embeds=...).logits."""
    _need_gpu()
    cfg = O.tiny_cfg()
    model, W = _model(cfg, seed=10)
    batch = O.make_synthetic_batch(cfg, 2, 40, 12, seed=10)
    ids, labels = batch["concatenated_input_ids"], batch["concatenated_labels"]
    images = torch.cat([batch["images"], batch["images"]])
    (_, _, _, _, emb, new_labels) = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, labels, images)
    logits = model(inputs_embeds=emb, labels=None).logits
    with torch.no_grad():
        feats = O.encode_images(images, W, cfg)
        emb_o, lab_o = O.prepare_inputs_labels_for_multimodal(ids, labels, feats, W["model.embed_tokens.weight"], cfg.model_max_length)
        logits_o = O.llama_logits(emb_o, W, cfg)
    assert torch.equal(new_labels.cpu(), lab_o)
    assert logits.shape == logits_o.shape
    lp, _ = O.get_batch_logps(logits.float().cpu(), lab_o)
    lp_o, _ = O.get_batch_logps(logits_o, lab_o)
    err = (logits.float().cpu() - logits_o).abs().max().item()
    print(f"forward().logits: max abs err {err:.3e} (logit scale {logits_o.abs().max().item():.2f}); seq log-probs {lp.tolist()} vs {lp_o.tolist()}")
    assert err <= 5e-2 and bool(((lp - lp_o).abs() <= 2e-3 * lp_o.abs()).all())
    with pytest.raises(NotImplementedError):
        model(inputs_embeds=emb, labels=new_labels)


def test_resume_restores_data_position(tmp_path):
    """A run stopped after 3 of 5 batches of epoch 0 resumes with batch 3 (not batch 0) and then reshuffles for epoch 1."""
    _need_gpu()
    cfg = O.tiny_cfg()
    from rlaif_v_amd.data import DataCollatorForDPODataset, SyntheticPreferenceDataset

    class Tok:
        pad_token_id = cfg.pad_token_id
    ds = SyntheticPreferenceDataset(n=10, vocab=cfg.vocab, text_len=36, prompt_len=12, image_size=cfg.image_size, seed=4)
    seen = []

    class Spy(DataCollatorForDPODataset):
        def __call__(self, instances):
            seen.append([int(i[0]["input_ids"][-3]) for i in instances])       # a fingerprint of the rows
            return super().__call__(instances)

    def make(max_steps):
        model, _ = _model(cfg, seed=11)
        tr = _trainer(model, max_steps=max_steps, per_device_train_batch_size=2, save_steps=3, logging_steps=100,
                      output_dir=str(tmp_path))
        tr.train_dataset, tr.data_collator = ds, Spy(Tok(), beta=0.1, mod_token_weight=1.0)
        return tr
    full = make(7)
    full.train()
    order_full = list(seen)
    seen.clear()
    part = make(3)
    part.train()
    assert part.state["batches_in_epoch"] == 3 and part.state["epoch"] == 0
    seen.clear()
    res = make(7)
    res.train(resume_from_checkpoint=str(tmp_path / "checkpoint-3"))
    assert res.state["global_step"] == 7
    assert seen == order_full[3:]                      # continues where it stopped; epoch 1 is a new permutation
    assert order_full[5:7] != order_full[0:2]
    assert torch.equal(res.model.store.flat_master, full.model.store.flat_master)


def test_evaluate_covers_the_tail_batch():
    """5 samples at eval batch 2 -> batches of 2, 2, 1: every sample counts once (sample-weighted mean), like the HF Trainer's
    evaluation loop; evaluation does not advance the LoRA dropout counter."""
    _need_gpu()
    cfg = O.tiny_cfg()
    model, W = _model(cfg, seed=12)
    from rlaif_v_amd.data import DataCollatorForDPODataset, SyntheticPreferenceDataset

    class Tok:
        pad_token_id = cfg.pad_token_id
    ds = SyntheticPreferenceDataset(n=5, vocab=cfg.vocab, text_len=40, prompt_len=12, image_size=cfg.image_size, seed=5)
    tr = _trainer(model, per_device_eval_batch_size=2)
    tr.eval_dataset, tr.data_collator = ds, DataCollatorForDPODataset(Tok(), beta=0.1, mod_token_weight=1.0)
    step0 = model._dropout_step
    m = tr.evaluate()
    assert model._dropout_step == step0
    tot = 0.0
    for idx in ([0, 1], [2, 3], [4]):
        batch = tr.data_collator([ds[i] for i in idx])
        with torch.no_grad():
            tot += len(idx) * float(O.dpo_step_forward(batch, W, cfg, sft_weight=0.0, dpo_weight=1.0)["loss"])
    ref = tot / 5
    assert abs(m["eval_loss"] - ref) <= 2e-3 * abs(ref), (m["eval_loss"], ref)


def test_checkpoint_refuses_another_parameter_layout(tmp_path, monkeypatch):
    """optimizer.pt carries a layout descriptor: a blob written with interleaved gate|up rows (RV_FUSE_SWIGLU=1) must not load
    into a block-layout store (same length, different row order), and an untagged legacy blob is of UNKNOWN layout: refused unless RV_CKPT_LEGACY_LAYOUT states it."""
    _need_gpu()
    cfg = O.tiny_cfg()
    model, _ = _model(cfg, seed=13)
    assert model.store.interleave_gu
    tr = _trainer(model, output_dir=str(tmp_path))
    tr.save_checkpoint(str(tmp_path / "ck"))
    blob = torch.load(str(tmp_path / "ck" / "optimizer.pt"), map_location="cpu")
    assert blob["layout"]["interleave_gu"] is True and blob["layout"]["n_train"] == model.store.n_train
    tr.load_checkpoint(str(tmp_path / "ck"))                                   # same layout: loads
    monkeypatch.setenv("RV_FUSE_SWIGLU", "0")
    other, _ = _model(cfg, seed=13)
    assert not other.store.interleave_gu
    with pytest.raises(ValueError, match="interleave_gu"):
        _trainer(other).load_checkpoint(str(tmp_path / "ck"))
    monkeypatch.delenv("RV_FUSE_SWIGLU")
    del blob["layout"]                                                          # a blob from before the descriptor existed
    os.makedirs(tmp_path / "old", exist_ok=True)
    torch.save(blob, str(tmp_path / "old" / "optimizer.pt"))
    with pytest.raises(ValueError, match="no layout descriptor"):              # unknown layout: never guessed
        tr.load_checkpoint(str(tmp_path / "old"))
    monkeypatch.setenv("RV_CKPT_LEGACY_LAYOUT", "block")                        # the caller states it; it must still match
    with pytest.raises(ValueError, match="interleave_gu"):
        tr.load_checkpoint(str(tmp_path / "old"))
    monkeypatch.setenv("RV_CKPT_LEGACY_LAYOUT", "interleaved")
    tr.load_checkpoint(str(tmp_path / "old"))
