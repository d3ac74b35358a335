#!/usr/bin/env python
"""Round-4 full-depth oracle fixtures, produced INSIDE the build container (8 vCPU, 62 GB) by the layer-streamed
evaluation of the oracle (oracle/streamed.py - the oracle's own functions, chain rule by hand at the layer boundaries):

    --base cfg1   fulldepth_cfg1_cond.pt (+ cfg1_step re-derived and cross-checked against the one-graph fixture that the
                  3 TB GPU-box host produced in round 3: two machines, two evaluation orders, one answer)
    --base cfg2   fulldepth_cfg2_step.pt, fulldepth_cfg2_cond.pt  (BASELINE config 2's packed shape, 2 pairs at L = 2048,
                  forward + backward + clip + AdamW; saturated and conditioned reference log-probs)

    --base cfg1m  fulldepth_cfg1m_step.pt  (round 5: config 1 on a batch whose 1e-3 loss bar has >= 4 sigma of margin)
    --base cfg5   fulldepth_cfg5_step.pt, fulldepth_cfg5_cond.pt  (round 5: BASELINE config 5, LoRA r = 64 at L = 4096, two pairs,
                  adapter + projector gradients, clip + AdamW)
    --base cfg5_drop  fulldepth_cfg5_drop.pt  (one pair at L = 4096, adapter dropout 0.05 with the device's masks replayed)
    --base cfg4   fulldepth_cfg4_step.pt, fulldepth_cfg4_cond.pt  (round 5: BASELINE config 4's trainable side - Resampler +
                  Mistral-7B-shaped decoder at L = 2048, forward_DPO + backward incl. resampler gradients)

Writes the fixtures to tests/golden/ and a log line per case to profiles/r0N_oracle_streamed_<base>.json.
Test infrastructure: imports oracle/ and tests/full_depth.py; nothing in the product path uses it.
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


def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", choices=["cfg1", "cfg1m", "cfg2", "cfg5", "cfg5_drop", "cfg4", "cfg1_outlier", "cfg1m_3step"], required=True)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--no-emulation", action="store_true")
    ap.add_argument("--workdir", default="/tmp/rv_oracle_multistep", help="cfg1m_3step: where the disk-backed g / m / v arrays live (81 GB)")
    ap.add_argument("--emu-backward", action="store_true", help="fresh run: also produce the bf16-emulated backward yardstick")
    ap.add_argument("--add-emu-backward", action="store_true",
                    help="load the base's existing fixtures and ADD the bf16-emulated oracle's backward (emu_grad_norms / emu_grad_cos: "
                         "the yardstick of the per-tensor gradient bars); the fp32 run is not repeated")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    base, cond = {"cfg1": ("cfg1_step", "cfg1_cond"), "cfg2": ("cfg2_step", "cfg2_cond"), "cfg5": ("cfg5_step", "cfg5_cond"),
                  "cfg5_drop": ("cfg5_drop_base", "cfg5_drop"), "cfg4": ("cfg4_step", "cfg4_cond"), "cfg1m": ("cfg1m_step", None),
                  "cfg1_outlier": ("cfg1_outlier", None), "cfg1m_3step": ("cfg1m_3step", None)}[args.base]
    cfg = FD.make_cfg(args.layers, base)
    t0 = time.time()
    W = FD.make_case_weights(base, cfg)
    log(f"weights ({args.layers} layers): {time.time() - t0:.0f} s")
    report = dict(host=dict(cpus=os.cpu_count(), threads=args.threads, torch=torch.__version__), layers=args.layers)
    if args.base == "cfg1m_3step":
        # round 6: T optimisation steps in the reference's mixed-precision arrangement (tests/full_depth.py oracle_multistep)
        suffix = "" if args.layers == 32 else f"_l{args.layers}"
        fx = FD.oracle_multistep(base, W, cfg, args.workdir, log=log)
        FD.save_fixture(fx, os.path.join(args.out, f"fulldepth_{base}{suffix}.pt"))
        report["steps"] = [dict(loss=s_["loss"], log_prob=s_["log_prob"].tolist(), grad_norm_total=s_["grad_norm_total"], clip_coef=s_["clip_coef"],
                                timings=s_["timings"]) for s_ in fx["steps"]]
        json.dump(report, open(os.path.join(REPO, "profiles", f"r06_oracle_streamed_{args.base}{suffix}.json"), "w"), indent=1)
        log("done")
        return
    if args.add_emu_backward:
        suffix = "" if args.layers == 32 else f"_l{args.layers}"
        names = [cs for cs in (base, cond) if os.path.exists(os.path.join(args.out, f"fulldepth_{cs}{suffix}.pt"))]
        fxs = {cs: torch.load(os.path.join(args.out, f"fulldepth_{cs}{suffix}.pt"), weights_only=False) for cs in names}
        worst = FD.add_emulated_backward(base, W, cfg, fxs, log=log)
        for cs, fx in fxs.items():
            FD.save_fixture(fx, os.path.join(args.out, f"fulldepth_{cs}{suffix}.pt"))
        path = os.path.join(REPO, "profiles", f"r05_oracle_emulated_backward_{args.base}{suffix}.json")
        json.dump(dict(report, emu_bf16_backward_worst_sample_cosine=worst, timings={cs: fxs[cs]["emu_backward_timings"] for cs in fxs}),
                  open(path, "w"), indent=1)
        log("done")
        return
    old = None
    emu_from = None
    old_path = os.path.join(REPO, "tests", "golden", f"fulldepth_{base}.pt")
    if args.base == "cfg1" and args.layers == 32 and os.path.exists(old_path):
        old = torch.load(old_path, weights_only=False)
        emu_from = old
    if args.no_emulation and emu_from is None:
        emu_from = dict(emu_per_token=None, emu_log_prob=None, emu_loss=None)
    fxs = FD.oracle_streamed(base, W, cfg, cond, emu_from=emu_from, own_refs=(args.base != "cfg5_drop"), log=log)
    if args.emu_backward:
        keep = {cs: fx for cs, fx in fxs.items() if not (cs == "cfg1_step" and old is not None)}
        report["emu_bf16_backward_worst_sample_cosine"] = FD.add_emulated_backward(base, W, cfg, keep, log=log)
    suffix = "" if args.layers == 32 else f"_l{args.layers}"
    os.makedirs(args.out, exist_ok=True)
    for cs, fx in fxs.items():
        if cs == "cfg1_step" and old is not None:
            # the one-graph run of round 3 (GPU-box host, 128 threads) vs this layer-streamed run (container, 8 threads)
            x = dict(loss=(old["loss"], fx["loss"]),
                     log_prob_max_rel=float(((old["log_prob"] - fx["log_prob"]).abs() / old["log_prob"].abs()).max()),
                     per_token_max_abs=float((old["per_token"] - fx["per_token"]).abs().max()),
                     grad_norm_total=(old["grad_norm_total"], fx["grad_norm_total"]),
                     grad_norm_worst_rel=max(abs(fx["grad_norms"][k] - n) / max(n, 1e-30) for k, n in old["grad_norms"].items() if n > 1e-9),
                     grad_sample_worst_cos=min(FD._cos(fx["grad_samples"][k], g) for k, g in old["grad_samples"].items()
                                               if float(g.norm()) > 0),
                     post_sample_max_abs=max(float((fx["post_samples"][k] - p).abs().max()) for k, p in old["post_samples"].items()))
            report["cfg1_step_streamed_vs_one_graph"] = x
            log("cfg1_step streamed vs one-graph fixture:", json.dumps(x))
            continue                                           # the committed one-graph fixture stays the reference
        FD.save_fixture(fx, os.path.join(args.out, f"fulldepth_{cs}{suffix}.pt"))
        if cs in FD.OUTLIER_STATS:
            fx["outlier_stats"] = report["outlier_stats"] = FD.OUTLIER_STATS[cs]
            FD.save_fixture(fx, os.path.join(args.out, f"fulldepth_{cs}{suffix}.pt"))
        report[cs] = dict(loss=fx["loss"], losses=fx["losses"].tolist(), log_prob=fx["log_prob"].tolist(), grad_norm_total=fx["grad_norm_total"],
                          clip_coef=fx["clip_coef"], timings=fx["timings"], beta_z=fx.get("beta_z"), n_variants=fx["n_variants"],
                          emu_s=fx.get("emu_s"))
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
    with open(os.path.join(REPO, "profiles", f"{'r04' if args.base in ('cfg1', 'cfg2') else 'r06' if args.base in ('cfg1_outlier',) else 'r05'}_oracle_streamed_{args.base}{suffix}.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    log("done")


if __name__ == "__main__":
    main()
