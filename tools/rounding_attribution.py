#!/usr/bin/env python
"""Attribution of the bf16 forward's per-token log-prob error to its rounding points, on the CPU.

Runs BASELINE config 1's batch (tests/full_depth.py ``cfg1_step``: 4 pairs, L = 1087, 32 layers, the seeded fp32 weights of the
committed fixture) forward-only through oracle/streamed.py with oracle/rounding.py's layer: fp32 arithmetic with bf16 roundings at
selectable groups of the HIP path's store points (R residual stream, N norm outputs, Q q/k/v (+ RoPE), P attention probabilities,
A attention output, G gate / up / act, F final hidden, V vision front).  Each variant's per-token log-probs are compared with the
fp32 oracle's (tests/golden/fulldepth_cfg1_step.pt): mean / RMS / max error, sequence-sum error, loss error and the one-sigma
size of rounding noise on the saturated loss (beta x rms x sqrt(n) of a pair, as tests/full_depth.py ``compare`` logs it).

    python tools/rounding_attribution.py [--variants all,-R,-N,...,+R] [--layers 32] [--case cfg1_step]

``all`` = every point on (the HIP path's arithmetic), ``-X`` = all but X, ``+X`` = only X, ``none`` = no rounding (must reproduce
the fixture).  ~3.5 minutes per variant on the 8-vCPU build container.  Writes profiles/r05_rounding_attribution.json.
Test infrastructure: imports oracle/ and tests/full_depth.py.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import torch  # noqa: E402

import full_depth as FD  # noqa: E402
from oracle import dpo_oracle as O  # noqa: E402
from oracle import rounding as RD  # noqa: E402
from oracle import streamed as S  # noqa: E402


def points_of(v: str) -> str:
    if v == "all":
        return RD.ALL_POINTS
    if v == "none":
        return ""
    if v.startswith("-"):
        return "".join(c for c in RD.ALL_POINTS if c not in v[1:])
    if v.startswith("+"):
        return v[1:]
    if v.startswith("="):          # explicit point string, e.g. =RNQPAGFoj (everything but the CLIP residual stream)
        return v[1:]
    raise SystemExit(f"bad variant {v!r}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="all,-R,-N,-Q,-P,-A,-G,-F,-V,+R,-RF")
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--case", default="cfg1_step")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r05_rounding_attribution.json"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    cfg = FD.make_cfg(args.layers, args.case)
    W = FD.make_case_weights(args.case, cfg)
    batch = FD.make_batch(args.case, cfg)
    fx = None
    path = os.path.join(REPO, "tests", "golden", f"fulldepth_{args.case}.pt")
    if args.layers == 32 and os.path.exists(path):
        fx = torch.load(path, weights_only=False)
    report = dict(case=args.case, layers=args.layers, points=dict(R="residual stream", N="RMSNorm outputs", Q="q/k/v outputs + RoPE",
                  P="attention probabilities (PV operand)", A="attention output", G="gate / up / act", F="final hidden", V="vision front"),
                  variants={})
    if os.path.exists(args.out):
        try:
            old = json.load(open(args.out))
            if old.get("case") == args.case and old.get("layers") == args.layers:
                report["variants"] = old["variants"]
        except ValueError:
            pass
    ref = None
    if fx is not None:
        ref = dict(per_token=fx["per_token"], log_prob=fx["log_prob"], loss=fx["loss"], labels=fx["labels"])
    for v in args.variants.split(","):
        pts = points_of(v)
        t0 = time.time()
        resolved = any(ch in pts for ch in RD.VISION_RESOLVED)
        front = RD.ResolvedLlavaFront(batch, cfg, W, pts) if resolved else RD.RoundedLlavaFront(batch, cfg, W, pts)
        res = S.dpo_step_streamed(batch, W, cfg, backward=False, layer_fn=RD.make_layer_fn(pts), hidden_fn=RD.make_hidden_fn(pts),
                                  front=front, row_chunk=2)
        mask = res["labels"][:, 1:] != O.IGNORE_INDEX
        cur = dict(per_token=res["per_token_logps"].float()[mask], log_prob=res["log_prob"].float(), loss=float(res["loss"]),
                   labels=res["labels"])
        if ref is None:            # no fixture (reduced depth): the first variant must be "none" and becomes the reference
            if pts != "":
                raise SystemExit("without a 32-layer fixture the first variant must be 'none'")
            ref = cur
        d = cur["per_token"] - ref["per_token"]
        B = cur["log_prob"].numel() // 2
        n_pair = (mask[:B].sum(1) + mask[B:].sum(1)).float()
        beta = float(batch["beta"])
        rms = float(d.pow(2).mean().sqrt())
        lp, lr = cur["log_prob"], ref["log_prob"]
        rec = dict(points=pts, per_token_mean_abs_err=float(d.abs().mean()), per_token_rms_err=rms, per_token_max_abs_err=float(d.abs().max()),
                   per_token_mean_signed_err=float(d.mean()),
                   seq_logp_max_rel_err=float(((lp - lr).abs() / lr.abs()).max()), loss=cur["loss"], loss_ref=float(ref["loss"]),
                   loss_rel_err=abs(cur["loss"] - float(ref["loss"])) / abs(float(ref["loss"])),
                   loss_one_sigma_rel=float(beta * rms * n_pair.sqrt().mean() / (B ** 0.5) / abs(float(ref["loss"]))),
                   logit_abs_err=(beta * ((lp[:B] - lp[B:]) - (lr[:B] - lr[B:]))).abs().tolist(), seconds=round(time.time() - t0, 1))
        report["variants"][v] = rec
        print(time.strftime("%H:%M:%S"), v, json.dumps(rec), flush=True)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
