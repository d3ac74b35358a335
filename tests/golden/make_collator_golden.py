"""Golden vectors for the batch-construction path, produced by the reference's own collator
(/root/reference muffin/train/train_muffin.py:37-112) on seeded synthetic instances.
Run in the build container:  python tests/golden/make_collator_golden.py"""
import os
import sys
import types

import torch
import transformers  # noqa: F401
import accelerate  # noqa: F401

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("wandb", types.ModuleType("wandb"))

from muffin.train.train_muffin import DataCollatorForDPODataset as RefCollator  # noqa: E402
from rlaif_v_amd.data import SyntheticPreferenceDataset  # noqa: E402


def instances(seed):
    ds = SyntheticPreferenceDataset(n=5, vocab=97, text_len=48, prompt_len=14, image_size=28, seed=seed, ragged=True)
    inst = [ds[i] for i in range(5)]
    # make chosen/rejected share long common spans so the difflib token-weight path is exercised
    g = torch.Generator().manual_seed(seed)
    for rej, win in inst:
        n = min(rej["input_ids"].numel(), win["input_ids"].numel())
        keep = torch.rand(n, generator=g) < 0.7
        keep[:14] = True
        rej["input_ids"][:n] = torch.where(keep, win["input_ids"][:n], rej["input_ids"][:n])
        rej["labels"] = rej["input_ids"].clone()
        rej["labels"][:14] = -100
        rej["ref_rej_per_token_logp"] = (-torch.rand(rej["input_ids"].numel() - 1, generator=g)).tolist()
        win["ref_win_per_token_logp"] = (-torch.rand(win["input_ids"].numel() - 1, generator=g)).tolist()
    return inst


if __name__ == "__main__":
    tok = types.SimpleNamespace(pad_token_id=0)
    out = {}
    for seed in (1, 2):
        batch = RefCollator(tok, beta=0.1, mod_token_weight=1.5)(instances(seed))
        out[seed] = {k: v for k, v in batch.items()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "collator.pt")
    torch.save(out, path)
    print({k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in out[1].items()})
    print("->", path, os.path.getsize(path))
