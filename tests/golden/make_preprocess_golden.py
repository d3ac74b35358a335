"""Golden vectors of the sample-encoding path from the reference's OWN functions (muffin/train/train_utils.py) over the
deterministic toy tokenizer.  Run in the build container: python tests/golden/make_preprocess_golden.py"""
import os
import sys
import types

import torch
import transformers  # noqa: F401
import accelerate  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("wandb", types.ModuleType("wandb"))

from functools import partial  # noqa: E402
from muffin.train.train_utils import encode_multimodal_preference_sample, preprocess_v1  # noqa: E402
from toy_tokenizer import SAMPLES, ToyTokenizer  # noqa: E402

if __name__ == "__main__":
    tok = ToyTokenizer()
    cfg = dict(image_processor=lambda img: torch.full((3, 4, 4), float(img)), keep_image_tag=True, is_multimodal=True,
               image_token_len=16, use_im_start_end=False)
    out = []
    for i, s in enumerate(SAMPLES):
        src = dict(image=i, question={"from": "human", "value": f"<image>\n{s['question']}"},
                   chosen={"from": "gpt", "value": s["chosen"]}, rejected={"from": "gpt", "value": s["rejected"]},
                   ref_win_logp=-1.0 - i, ref_rej_logp=-2.0 - i, ref_win_avg_logp=-0.1, ref_rej_avg_logp=-0.2,
                   ref_win_per_token_logp=[0.0, -1.0], ref_rej_per_token_logp=[-2.0])
        rej, win = encode_multimodal_preference_sample(src, tok, cfg, preprocess_func=partial(preprocess_v1, has_image=True))
        out.append(dict(rej=rej, win=win))
    torch.save(out, os.path.join(HERE, "preprocess.pt"))
    print(out[0]["win"]["input_ids"].tolist(), out[0]["win"]["labels"].tolist())
