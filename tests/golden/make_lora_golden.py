"""Pins the oracle's LoRA arithmetic against peft ITSELF - when the wheel is importable.

The reference wraps the language model with peft 0.10 (muffin/train/train_llava15_lora.py:304-318: LoraConfig(r, lora_alpha,
target_modules=find_all_linear_names(model), lora_dropout, bias, task_type="CAUSAL_LM") + get_peft_model; merge path
llava/model/builder.py:81-85).  peft is NOT installed in the build image, so today `oracle.dpo_oracle.lora_linear` is a
restatement of peft's published `lora.Linear.forward` anchored on identities (tests/test_lora_oracle.py) - "parity unpinned".
The day `import peft` succeeds, run

    python tests/golden/make_lora_golden.py

and commit tests/golden/tiny_lora_peft.pt: tests/test_lora_oracle.py::test_oracle_matches_peft_golden then replays it (it is
skipped while the fixture does not exist).  Build container only (needs /root/reference)."""
import importlib.util
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
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
