"""Golden vectors of the reference's log-prob reductions on seeded random logits: get_batch_logps,
get_batch_logps_minicpm (muffin/eval/muffin_inference_logp.py:21-115) and compute_weighted_logp
(muffin/train/trainers.py:128-137).  Run in the build container:  python tests/golden/make_logps_golden.py"""
import os
import sys
import types

import torch
import transformers  # noqa: F401
import accelerate  # noqa: F401
from transformers import Trainer  # noqa: F401

sys.modules.setdefault("wandb", types.ModuleType("wandb"))
sys.path.insert(0, "/root/reference")
from muffin.eval.muffin_inference_logp import get_batch_logps, get_batch_logps_minicpm  # noqa: E402
from muffin.train.trainers import compute_weighted_logp  # noqa: E402


def inputs(seed=0):
    g = torch.Generator().manual_seed(seed)
    S, L, V = 4, 23, 50
    logits = torch.randn(S, L, V, generator=g) * 3
    labels = torch.randint(0, V, (S, L), generator=g)
    labels[:, :5] = -100
    labels[1, 15:] = -100
    labels[3] = -100                      # a row without targets: average = 0/0 = NaN
    weight = torch.where(torch.rand(S, L - 1, generator=g) < 0.3, torch.tensor(3.0), torch.tensor(1.0))
    return logits, labels, weight


if __name__ == "__main__":
    logits, labels, weight = inputs()
    pt, lp, avg = get_batch_logps(logits, labels, return_all=True)
    ptm, lpm, avgm = get_batch_logps_minicpm(logits, labels, return_all=True)
    out = dict(per_token=pt, log_prob=lp, avg=avg, per_token_minicpm=ptm, log_prob_minicpm=lpm, avg_minicpm=avgm,
               weighted_sum=compute_weighted_logp(pt, labels, weight, False),
               weighted_avg=compute_weighted_logp(pt, labels, weight, True))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "logps_fns.pt")
    torch.save(out, path)
    print("wrote", path, {k: tuple(v.shape) for k, v in out.items()})
