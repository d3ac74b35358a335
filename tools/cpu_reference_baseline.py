"""CPU baseline of BASELINE.md section 2: the REFERENCE'S OWN step timed on the host cores of the build container.

What is timed (BASELINE config 1: 4 synthetic 336-px pairs, text length T = 512 -> spliced length L = 1087, fp32, 1 step):
    get_beta_and_logps(..., is_llava15=True)  ->  dpo_loss  ->  loss mix (muffin/train/trainers.py:289-301)  ->  backward()
    ->  clip_grad_norm_(1.0)  ->  torch.optim.AdamW (HF adamw_torch defaults, decay on matrices only)
executed from /root/reference (read-only) on top of the installed transformers 5.15 Llama / CLIP.  The HF ``Trainer``
wrapper itself cannot run unmodified on transformers 5.15, so the loop around the reference
functions is this <=30-line shim.  Full-depth fp32 training state (108 GB) does not fit the container's 62 GB: the model is
built at full WIDTH (d 4096, f 11008, V 32000, CLIP-L/14-336) with 2 and 4 language-model layers and the per-layer slope is
extrapolated to 32 layers.

    python tools/cpu_reference_baseline.py [--pairs 4] [--text-len 512] > profiles/r02_cpu_reference_baseline.log

Runs in the build container only (needs /root/reference); nothing on the GPU box imports this file.
"""
import argparse
import importlib.util
import json
import os
import sys
import time
import types

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4)
    ap.add_argument("--text-len", type=int, default=512)
    ap.add_argument("--depths", type=str, default="2,4")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    spec = importlib.util.spec_from_file_location("_mk", os.path.join(REPO, "tests", "golden", "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)             # imports the reference (wandb stub, transformers first)
    O = mk.O
    from muffin.train.trainers import get_beta_and_logps, dpo_loss

    res = {}
    for depth in [int(x) for x in a.depths.split(",")]:
        cfg = O.LlavaCfg(layers=depth, model_max_length=2048)
        W = O.make_weights(cfg, seed=0, bf16_round=False)
        model = mk.build_reference_model(cfg, W)
        del W
        decay = [p for n, p in model.named_parameters() if p.requires_grad and not (n.endswith("bias") or "norm" in n)]
        nodecay = [p for n, p in model.named_parameters() if p.requires_grad and (n.endswith("bias") or "norm" in n)]
        opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}],
                                lr=5e-7, betas=(0.9, 0.999), eps=1e-8)
        batch = O.make_synthetic_batch(cfg, a.pairs, a.text_len, 64, seed=0, ragged=False)
        args = types.SimpleNamespace(dpo_use_average=False, task="DPO", dpo_token_weighted=False, past_index=-1)
        times = []
        for it in range(3):                 # first iteration = warm-up (allocator, oneDNN primitive caches)
            data = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
            t0 = time.perf_counter()
            pw, pr, rw, rr, beta = get_beta_and_logps(data, model, args, is_llava15=True)
            losses, cw, cr = dpo_loss(pw, pr, rw, rr, beta=beta)
            loss = 1.0 * losses.mean() - 0.0 * pw.mean()
            t1 = time.perf_counter()
            loss.backward()
            t2 = time.perf_counter()
            torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            t3 = time.perf_counter()
            times.append((t1 - t0, t2 - t1, t3 - t2))
            print(f"depth {depth} iter {it}: fwd {t1 - t0:.2f}s bwd {t2 - t1:.2f}s clip+AdamW {t3 - t2:.2f}s loss {float(loss):.6f}",
                  flush=True)
        res[depth] = tuple(min(t[i] for t in times[1:]) for i in range(3))      # best of the timed iterations
        del model, opt
    ds = sorted(res)
    lo, hi = ds[0], ds[-1]
    slope = [(res[hi][i] - res[lo][i]) / (hi - lo) for i in range(3)]
    fixed = [res[lo][i] - lo * slope[i] for i in range(3)]
    full = [fixed[i] + 32 * slope[i] for i in range(3)]
    step = sum(full)
    out = dict(kind="reference", cores=a.threads, cpu=open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
               pairs=a.pairs, text_len=a.text_len, spliced_len=a.text_len - 1 + 576, dtype="fp32",
               measured_s={d: dict(fwd=res[d][0], bwd=res[d][1], opt=res[d][2]) for d in ds},
               per_layer_s=dict(fwd=slope[0], bwd=slope[1], opt=slope[2]), fixed_s=dict(fwd=fixed[0], bwd=fixed[1], opt=fixed[2]),
               extrapolated_32_layers_s=dict(fwd=full[0], bwd=full[1], opt=full[2], step=step),
               pairs_per_s=a.pairs / step)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
