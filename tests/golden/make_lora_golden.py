"""Pins the oracle's LoRA arithmetic against peft ITSELF - when the wheel is importable.

The reference wraps the language model with peft 0.10 (muffin/train/train_llava15_lora.py:304-318: LoraConfig(r, lora_alpha,
target_modules=find_all_linear_names(model), lora_dropout, bias, task_type="CAUSAL_LM") + get_peft_model; merge path
llava/model/builder.py:81-85).  peft is NOT installed in the build image, so today `oracle.dpo_oracle.lora_linear` is a
restatement of peft's published `lora.Linear.forward` anchored on identities (tests/test_lora_oracle.py) - "parity unpinned".
The day `import peft` succeeds, run

    python tests/golden/make_lora_golden.py

and commit tests/golden/tiny_lora_peft.pt: tests/test_lora_oracle.py::test_oracle_matches_peft_golden then replays it (it is
skipped while the fixture does not exist).  Build container only (needs /root/reference).

    python tests/golden/make_lora_golden.py --merged        (round 4: works WITHOUT peft)

pins the adapter arithmetic against outputs of the REFERENCE ITSELF through the state peft's own merge leaves behind:
``merge_and_unload`` (llava/model/builder.py:81-85) replaces every wrapped nn.Linear weight by W' = W + (alpha/r) B A and
drops the adapter, so the reference's ``LlavaLlamaForCausalLM`` run on W' IS the reference output of the adapter model's
forward (dropout off), and its weight gradients dW' give the adapter gradients by the chain rule of W'(A, B):

    dA = (alpha/r) B^T dW'          dB = (alpha/r) dW' A^T

Cases: tiny (r 16), tiny grouped-query (r 16), LLaVA-1.5-7B widths at 2 layers (r 64, alpha 16: the shipped script's
values, train_llava15_lora.py:113-114).  Writes tests/golden/lora_merged_*.pt: log-probs, loss, every dW' norm, the
chain-rule adapter gradients (norms, sampled elements; in full at tiny size) and the projector gradients (the LoRA run
trains the projector too: llava_arch.py:90-93).  Replayed on the oracle (tests/test_lora_oracle.py) and on the HIP LoRA
path with dropout 0 (tests/test_lora_gpu.py)."""
import importlib.util
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


N_SAMPLE = 512


def sample_index(name: str, numel: int, n: int = N_SAMPLE) -> torch.Tensor:
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    return torch.randint(0, numel, (min(n, numel),), generator=g)


def merged_case(mk, name, cfg, r, alpha, n_pairs, text_len, prompt_len, seed, keep_full):
    """Reference model on W' = W + (alpha/r) B A; adapter gradients by the chain rule (module docstring)."""
    O = mk.O
    from muffin.train.trainers import get_beta_and_logps, dpo_loss
    from muffin.eval.muffin_inference_logp import get_batch_logps
    scale = alpha / r
    W = O.make_weights(cfg, seed=seed)
    lw = O.make_lora_weights(cfg, r, seed=seed + 1, b_std=0.02)
    Wm = {k: v.clone() for k, v in W.items()}
    targets = sorted({k[:-len(".lora_A.weight")] for k in lw if k.endswith(".lora_A.weight")})
    for t in targets:                                   # what merge_and_unload leaves in the module: W + scale * B @ A
        A, B = lw[t + ".lora_A.weight"].double(), lw[t + ".lora_B.weight"].double()
        Wm[t + ".weight"] = (W[t + ".weight"].double() + scale * (B @ A)).float()
    model = mk.build_reference_model(cfg, Wm)
    batch = O.make_synthetic_batch(cfg, n_pairs, text_len, prompt_len, seed=seed)
    args = types.SimpleNamespace(dpo_use_average=False, task="DPO", dpo_token_weighted=False, past_index=-1)
    with torch.no_grad():
        cat_images = torch.cat([batch["images"], batch["images"]], 0)
        (_, _, _, _, emb, lab) = model.prepare_inputs_labels_for_multimodal(
            input_ids=batch["concatenated_input_ids"], position_ids=None, attention_mask=None,
            past_key_values=None, labels=batch["concatenated_labels"], images=cat_images)
        logits = model.forward(inputs_embeds=emb, labels=None).logits
        per_tok, lp, _ = get_batch_logps(logits, lab, return_all=True)
        del logits
    data = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    pw, pr, rw, rr, beta = get_beta_and_logps(data, model, args, is_llava15=True)
    losses, _, _ = dpo_loss(pw, pr, rw, rr, beta=beta)
    loss = losses.mean()
    loss.backward()
    grads = {k: p.grad.detach() for k, p in model.named_parameters() if p.grad is not None}
    gnorm, gsamp, gfull, dwnorm = {}, {}, {}, {}
    for t in targets:
        dW = grads[t + ".weight"].double()
        dwnorm[t] = float(dW.norm())
        A, B = lw[t + ".lora_A.weight"].double(), lw[t + ".lora_B.weight"].double()
        for key, g in ((t + ".lora_A.weight", scale * (B.t() @ dW)), (t + ".lora_B.weight", scale * (dW @ A.t()))):
            gnorm[key] = float(g.norm())
            gsamp[key] = g.flatten()[sample_index(key, g.numel())].float()
            if keep_full:
                gfull[key] = g.float()
    for k, g in grads.items():
        if "mm_projector" in k:
            gnorm[k] = float(g.double().norm())
            gsamp[k] = g.flatten()[sample_index(k, g.numel())].float().clone()
    out = dict(cfg=O.asdict(cfg), r=r, lora_alpha=alpha, seed=seed, n_pairs=n_pairs, text_len=text_len, prompt_len=prompt_len,
               labels=lab, per_token_logps=per_tok, log_prob=lp, policy_win_logp=pw.detach(), policy_rej_logp=pr.detach(),
               losses=losses.detach(), loss=loss.detach(), merged_weight_grad_norms=dwnorm, grad_norms=gnorm, grad_samples=gsamp,
               grad_full=gfull, how="reference LlavaLlamaForCausalLM on merge_and_unload weights; adapter gradients by the chain rule")
    path = os.path.join(HERE, f"{name}.pt")
    torch.save(out, path)
    print(name, "loss", float(loss), "logp", lp.tolist(), "->", path, os.path.getsize(path), "bytes", flush=True)


def main_merged():
    spec = importlib.util.spec_from_file_location("_mk", os.path.join(HERE, "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    O = mk.O
    torch.set_num_threads(8)
    merged_case(mk, "lora_merged_tiny", O.tiny_cfg(), 16, 16, 2, 40, 12, seed=5, keep_full=True)
    merged_case(mk, "lora_merged_tiny_gqa", O.tiny_gqa_cfg(), 16, 32, 2, 40, 12, seed=6, keep_full=True)
    if "--no-full-width" not in sys.argv:
        merged_case(mk, "lora_merged_fullwidth_l2", mk.fullwidth_cfg(), 64, 16, 2, 96, 40, seed=22, keep_full=False)
    return 0


def main():
    if "--merged" in sys.argv:
        return main_merged()
    try:
        import peft
    except ImportError:
        print("peft is not importable in this image: nothing generated (the LoRA row stays 'parity unpinned' against peft)")
        return 0
    spec = importlib.util.spec_from_file_location("_mk", os.path.join(HERE, "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    O = mk.O
    from muffin.train.trainers import get_beta_and_logps, dpo_loss
    cfg, r, alpha, seed = O.tiny_cfg(), 16, 16, 5
    W = O.make_weights(cfg, seed=seed)
    model = mk.build_reference_model(cfg, {k: v.clone() for k, v in W.items()})
    targets = sorted({n.split(".")[-1] for n, m in model.named_modules()
                      if isinstance(m, torch.nn.Linear) and "mm_projector" not in n and "vision_tower" not in n and "lm_head" not in n})
    pcfg = peft.LoraConfig(r=r, lora_alpha=alpha, target_modules=targets, lora_dropout=0.0, bias="none", task_type="CAUSAL_LM")
    pm = peft.get_peft_model(model, pcfg)
    lw = O.make_lora_weights(cfg, r, seed=seed + 1, b_std=0.02)
    sd = pm.state_dict()
    for k, v in lw.items():
        key = "base_model.model." + k.replace(".lora_A.weight", ".lora_A.default.weight").replace(".lora_B.weight", ".lora_B.default.weight")
        assert key in sd, key
        sd[key].copy_(v)
    for n, p in pm.named_parameters():                       # llava_arch.py:90-93: the projector is re-enabled under LoRA
        if "mm_projector" in n:
            p.requires_grad_(True)
    batch = O.make_synthetic_batch(cfg, 2, 40, 12, seed=seed)
    args = types.SimpleNamespace(dpo_use_average=False, task="DPO", dpo_token_weighted=False, past_index=-1)
    data = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    pw, pr, rw, rr, beta = get_beta_and_logps(data, pm, args, is_llava15=True)
    losses, cw, cr = dpo_loss(pw, pr, rw, rr, beta=beta)
    loss = losses.mean()
    loss.backward()
    grads = {n.replace("base_model.model.", "").replace(".default", ""): p.grad.detach().clone()
             for n, p in pm.named_parameters() if p.grad is not None}
    out = dict(cfg=O.asdict(cfg), r=r, lora_alpha=alpha, seed=seed, n_pairs=2, text_len=40, prompt_len=12,
               policy_win_logp=pw.detach(), policy_rej_logp=pr.detach(), loss=loss.detach(),
               grad_norms={k: float(g.double().norm()) for k, g in grads.items()}, peft_version=peft.__version__)
    path = os.path.join(HERE, "tiny_lora_peft.pt")
    torch.save(out, path)
    print("wrote", path, "loss", float(loss))
    return 0


if __name__ == "__main__":
    sys.exit(main())
