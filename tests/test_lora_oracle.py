"""CPU checks of the LoRA restatement in the oracle.

peft is absent offline, so the adapter arithmetic cannot be pinned against the reference directly; it is pinned
through identities against the base path, which IS pinned against the reference's own outputs
(tests/golden/tiny_*.pt):  (1) peft's initial state (lora_B = 0) is the identity, (2) an adapter model equals the
base model run on merged weights W + (alpha/r) B A (merge_and_unload, llava/model/builder.py:81-85), (3) autograd
gradients of the adapters agree with the closed forms the HIP path implements."""
import os

import torch

from oracle import dpo_oracle as O


def _setup(r=16, b_std=0.02):
    cfg = O.tiny_cfg()
    W = O.make_weights(cfg, seed=3)
    W.update(O.make_lora_weights(cfg, r, seed=4, b_std=b_std))
    batch = O.make_synthetic_batch(cfg, 2, 40, 12, seed=9)
    return cfg, W, batch


def test_peft_initial_state_is_identity(golden_dir):
    g = torch.load(os.path.join(golden_dir, "tiny_b2.pt"), weights_only=False)
    cfg = O.LlavaCfg(**g["cfg"])
    W = O.make_weights(cfg, seed=g["seed"])
    W.update(O.make_lora_weights(cfg, 16, b_std=None))             # lora_B = 0
    batch = O.make_synthetic_batch(cfg, g["n_pairs"], g["text_len"], g["prompt_len"], seed=g["seed"])
    out = O.dpo_step_forward(batch, W, cfg, sft_weight=g["sft_weight"], dpo_weight=1.0, lora_scale=16 / 16)
    torch.testing.assert_close(out["log_prob"], g["log_prob"], rtol=1e-4, atol=1e-3)    # the reference's own numbers
    torch.testing.assert_close(out["loss"], g["loss"], rtol=1e-4, atol=1e-4)


def test_adapter_equals_merged_weights():
    cfg, W, batch = _setup()
    scale = 16 / 16
    a = O.dpo_step_forward(batch, W, cfg, sft_weight=0.1, dpo_weight=1.0, lora_scale=scale)
    m = O.dpo_step_forward(batch, O.merge_lora(W, scale), cfg, sft_weight=0.1, dpo_weight=1.0)
    base = O.dpo_step_forward(batch, W, cfg, sft_weight=0.1, dpo_weight=1.0)       # adapters ignored
    torch.testing.assert_close(a["per_token_logps"], m["per_token_logps"], rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(a["loss"], m["loss"], rtol=1e-4, atol=1e-5)
    assert (a["log_prob"] - base["log_prob"]).abs().max() > 1e-2       # the adapter really changes the output


def test_adapter_gradients_closed_form():
    """dA = s (dy B)^T x, dB = dy^T (s x A^T) for one projection, against autograd through the whole step."""
    cfg, W, batch = _setup()
    scale = 0.25
    name = "model.layers.1.mlp.down_proj"
    for k in O.lora_trainable_names(W):          # everything the LoRA run trains, so activations carry gradients
        W[k].requires_grad_(True)
    captured = {}
    orig = O.F.linear

    def spy(x, w, b=None):
        y = orig(x, w, b)
        if w is W[name + ".weight"]:
            captured["x"] = x.detach()
            y.retain_grad()
            captured["y"] = y
        return y
    O.F.linear = spy
    try:
        out = O.dpo_step_forward(batch, W, cfg, sft_weight=0.0, dpo_weight=1.0, lora_scale=scale)
        out["loss"].backward()
    finally:
        O.F.linear = orig
    x = captured["x"].reshape(-1, cfg.ffn)
    dy = captured["y"].grad.reshape(-1, cfg.hidden)        # d loss / d (base + adapter) output: same tensor shape/grad
    A, B = W[name + ".lora_A.weight"].detach(), W[name + ".lora_B.weight"].detach()
    dA = scale * (dy @ B).t() @ x
    dB = dy.t() @ (scale * x @ A.t())
    torch.testing.assert_close(W[name + ".lora_A.weight"].grad, dA, rtol=1e-3, atol=1e-7)
    torch.testing.assert_close(W[name + ".lora_B.weight"].grad, dB, rtol=1e-3, atol=1e-7)


def test_lora_train_step_touches_only_adapters_and_projector():
    cfg, W, batch = _setup()
    before = {k: v.clone() for k, v in W.items()}
    state = {}
    O.dpo_train_step(batch, W, cfg, state, lr=1e-3, step=1, sft_weight=0.0, dpo_weight=1.0, lora_scale=1.0)
    moved = {k for k in W if not torch.equal(W[k], before[k])}
    assert moved and all(".lora_" in k or "mm_projector" in k for k in moved), sorted(moved)[:5]
    assert any(".lora_A." in k for k in moved) and any(".lora_B." in k for k in moved) and any("mm_projector" in k for k in moved)
