#!/usr/bin/env python
"""ONE full DPO optimisation step of BASELINE config 1 at all 32 layers (and config 2's forward) - HIP path vs the fp32
CPU oracle on the SAME weights and batch, on the GPU box (its host has 3 TB of RAM and 128 cores; the build container
cannot hold the 7B fp32 training state).  Writes

    gpurun_out/fulldepth_cfg1_step.pt, fulldepth_cfg2_fwd.pt   the oracle's outputs -> committed under tests/golden/
    gpurun_out/r03_parity_full_depth.json                      measured parity numbers + the MEASURED full-depth CPU step
                                                               time (bench.py's cpu_baseline quotes it) -> profiles/

    python tools/full_depth_parity.py [--layers 32] [--threads 128] [--no-hip] [--no-emulation]

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
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--threads", type=int, default=min(128, os.cpu_count() or 8))
    ap.add_argument("--no-hip", action="store_true", help="oracle + fixtures only (dry runs without a GPU)")
    ap.add_argument("--no-emulation", action="store_true")
    ap.add_argument("--cases", default="cfg2_fwd,cfg1_step")
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(args.threads)
    cfg = FD.make_cfg(args.layers)
    t0 = time.time()
    W = O.make_weights(cfg, seed=FD.WEIGHT_SEED)
    log(f"weights ({args.layers} layers): {time.time() - t0:.0f} s")
    cases = [c for c in args.cases.split(",") if c]
    hip = {}
    if not args.no_hip:
        model, trainer = FD.build_model(cfg, W, with_optimizer=True)
        for case in cases:                       # cfg2_fwd first: cfg1_step moves the weights
            t0 = time.time()
            hip[case] = FD.hip_case(case, model, trainer, cfg, full_grads=True)
            torch.cuda.synchronize()
            log(f"[{case}] HIP path: {time.time() - t0:.1f} s (first call, includes warm-up), loss {hip[case]['loss']:.6f}")
        del model, trainer
        torch.cuda.empty_cache()
        open(os.path.join(args.out, ".hip_done"), "w").write("done\n")        # the GPU is free from here on
    report = dict(host=dict(cpus=os.cpu_count(), threads=args.threads, torch=torch.__version__), layers=args.layers)
    W0 = {k: v.clone() for k, v in W.items() if not k.startswith(O.VT)} if "cfg1_step" in cases else None
    for case in cases:
        fx = FD.oracle_case(case, W, cfg, emulate=not args.no_emulation, log=log)
        FD.save_fixture(fx, os.path.join(args.out, f"fulldepth_{case}.pt"))
        if case in hip:
            try:
                m = FD.compare(case, hip[case], fx, W0=W0 if FD.CASES[case]["step"] else None, check=True)
                m["bars_met"] = True
            except AssertionError as e:
                m = FD.compare(case, hip[case], fx, W0=W0 if FD.CASES[case]["step"] else None, check=False)
                m["bars_met"], m["failed_bar"] = False, repr(e)[:300]
            report[case] = m
        else:
            report[case] = dict(loss_oracle=fx["loss"], seq_logp_oracle=fx["log_prob"].tolist())
        if "timings" in fx:
            t = fx["timings"]
            step_s = t["fwd_s"] + t["bwd_s"] + t["opt_s"]
            report["cpu_step_measured"] = dict(kind="port, full depth, measured", pairs=FD.CASES[case]["pairs"], layers=args.layers,
                                               fwd_s=t["fwd_s"], bwd_s=t["bwd_s"], opt_s=t["opt_s"], step_s=step_s,
                                               pairs_per_s=FD.CASES[case]["pairs"] / step_s, threads=args.threads,
                                               host_cpus=os.cpu_count(), dtype="fp32",
                                               what="oracle.dpo_train_step: forward + autograd backward + clip_grad_norm_ + AdamW, "
                                                    "BASELINE config 1 (4 pairs, T=512 -> L=1087)")
        for k in ("emu_s", "fwd_s"):
            if k in fx:
                report.setdefault(case, {})[f"oracle_{k}"] = fx[k]
        log(json.dumps({k: v for k, v in report[case].items() if not isinstance(v, (list, dict))}))
        with open(os.path.join(args.out, "r03_parity_full_depth.json"), "w") as fh:
            json.dump(report, fh, indent=1)
        fx.pop("_full_grads", None)
        hip.get(case, {}).pop("_full_grads", None)
    log("done")


if __name__ == "__main__":
    main()
