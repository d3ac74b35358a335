"""Golden vectors of the OmniLMM sample encoding from the reference's OWN omni_preprocess (omnilmm/train/train_utils.py:50-151)
and chat.py's wrap_question_for_omni_lmm over the deterministic toy tokenizer.
Run in the build container: python tests/golden/make_omni_preprocess_golden.py"""
import copy
import os
import sys
import types
import warnings

import torch
import transformers  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
for name in ("wandb", "cv2"):
    sys.modules.setdefault(name, types.ModuleType(name))

from omnilmm.train.train_utils import omni_preprocess  # noqa: E402
from toy_tokenizer import OMNI_CONVERSATIONS, OmniToyTokenizer  # noqa: E402

if __name__ == "__main__":
    tok = OmniToyTokenizer()
    out = {"train": [], "generation": []}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for conv in OMNI_CONVERSATIONS:
            for mode, gen in (("train", False), ("generation", True)):
                d = omni_preprocess(sources=[copy.deepcopy(conv)], tokenizer=tok, generation=gen)
                out[mode].append(dict(input_ids=d["input_ids"][0].clone(), labels=d["labels"][0].clone()))
    torch.save(out, os.path.join(HERE, "omni_preprocess.pt"))
    for r in out["train"]:
        print(r["input_ids"].tolist(), r["labels"].tolist())
