#!/usr/bin/env python
"""BASELINE config 1's literal batch (cfg1_step: the 1e-3 loss bar is a 0.2-sigma statement on it) under the numerics switches of round
6: residual stream bf16 / fp32 (RV_RESID_FP32) x RoPE in the q|k|v epilogue on / off (RV_FUSE_ROPE_FWD); SwiGLU always from the fp32
accumulators.  Forward only, 32 layers, against the committed fixture.  Usage (GPU box): python tools/exp_cfg1_step_numerics.py"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: E402
import full_depth as FD  # noqa: E402
from oracle import dpo_oracle as O  # noqa: E402

cfg = FD.make_cfg(32)
W = O.make_weights(cfg, seed=FD.WEIGHT_SEED)
model, trainer = FD.build_model(cfg, W, with_optimizer=False)
out = {}
for case in ("cfg1_step", "cfg1m_step", "cfg2_step"):
    fx = torch.load(os.path.join(REPO, "tests", "golden", f"fulldepth_{case}.pt"), weights_only=False)
    batch = FD.make_batch(case, cfg)
    mask = fx["labels"][:, 1:] != -100
    for resid in (False, True):
        for rope in (False, True):
            model.resid_fp32, model.fuse_rope_fwd = resid, rope
            model.train(False)
            loss = float(trainer.compute_loss(model, dict(batch)))
            o = model.last_out
            sd = o.per_token_logp.float().cpu() - fx["per_token"]
            r = dict(loss=loss, loss_oracle=fx["loss"], loss_rel_err=abs(loss - fx["loss"]) / abs(fx["loss"]),
                     per_token_rms_err=float(sd.pow(2).mean().sqrt()), per_token_mean_abs_err=float(sd.abs().mean()),
                     emu_per_token_rms=float((fx["emu_per_token"] - fx["per_token"]).pow(2).mean().sqrt()))
            out[f"{case} resid_fp32={int(resid)} rope_fused={int(rope)}"] = r
            print(case, f"resid_fp32={int(resid)} rope_fused={int(rope)}", json.dumps(r), flush=True)
os.makedirs(os.path.join(REPO, "gpurun_out", "r06"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "r06", "cfg1_step_numerics.json"), "w"), indent=1)
