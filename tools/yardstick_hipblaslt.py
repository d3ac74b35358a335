#!/usr/bin/env python
"""Same-box yardstick for the step's MFMA GEMMs: this library's NN / TN kernels against hipBLASLt
(through torch.matmul - a library GEMM used ONLY here, as a measuring stick; the product path never calls it) on the six GEMM
shapes of the LLaVA-1.5-7B step at the bench's 27,664 packed rows, in ONE process, alternating A / B / A / B in windows of
~2 s with package power and shader clock sampled (rocm-smi) inside every window.  Under the 1400 W package cap the quantity
that separates two correct GEMM kernels is joules per flop, which shows as the clock each sustains at equal watts.

    python tools/yardstick_hipblaslt.py [--seconds 2.0] [--rounds 2] > gpurun_out/yardstick.log

Prints one line per (shape, implementation, round) and a JSON summary (median round) at the end.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from rlaif_v_amd import ops  # noqa: E402

BF = torch.bfloat16


def sampler(stop, rows):
    while not stop.is_set():
        try:
            d = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True,
                                          timeout=10).stdout).get("card0", {})
            clk = d.get("sclk clock speed:")
            w = d.get("Current Socket Graphics Package Power (W)")
            rows.append((float(str(clk).strip("()Mhz ")) if clk else None, float(w) if w else None))
        except Exception:      # noqa: BLE001
            pass
        time.sleep(0.05)


def window(fn, seconds):
    rows, stop = [], threading.Event()
    th = threading.Thread(target=sampler, args=(stop, rows))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    th.start()
    t0, n = time.time(), 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    while time.time() - t0 < seconds:
        for _ in range(10):
            fn()
        n += 10
        torch.cuda.synchronize()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    stop.set()
    th.join()
    rows = [r for r in rows[2:] if r[0] and r[1]]
    clk = sorted(r[0] for r in rows)[len(rows) // 2] if rows else None
    watts = sorted(r[1] for r in rows)[len(rows) // 2] if rows else None
    return ms, clk, watts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--rows", type=int, default=27664)
    args = ap.parse_args()
    R, d, f = args.rows, 4096, 11008
    g = torch.Generator(device="cuda").manual_seed(0)

    def rn(*shape, scale=1.0):
        return (torch.randn(*shape, device="cuda", generator=g) * scale).to(BF)

    # (name, form, M, N, K): NN = activation [M, K] x weight^T-copy [K, N]; TN = dY^T [K, M]^T x X [K, N] (K = tokens)
    shapes = [("qkv fwd / d(qkv) dgrad", "nn", R, 3 * d, d), ("o fwd / d(o) dgrad", "nn", R, d, d), ("gate|up fwd", "nn", R, 2 * f, d),
              ("down fwd", "nn", R, d, f), ("d(gate|up) dgrad", "nn", R, d, 2 * f),
              ("wqkv wgrad", "tn", 3 * d, d, R), ("wgu wgrad", "tn", 2 * f, d, R), ("wdown wgrad", "tn", d, f, R)]
    summary = {}
    for name, form, M, N, K in shapes:
        if form == "nn":
            a, b = rn(M, K), rn(K, N, scale=0.02)
            out = torch.empty(M, N, device="cuda", dtype=BF)
            ours = lambda: ops.gemm_nn(a, b, out=out)                # noqa: E731
            lib = lambda: torch.matmul(a, b, out=out)                # noqa: E731
        else:
            p, q = rn(K, M), rn(K, N)
            out = torch.empty(M, N, device="cuda", dtype=BF)
            ours = lambda: ops.gemm_tn(p, q, out=out)                # noqa: E731
            pt = p.t()
            lib = lambda: torch.matmul(pt, q, out=out)               # noqa: E731
        fl = 2.0 * M * N * K
        rec = {"ours": [], "hipblaslt": []}
        for r in range(args.rounds):
            for impl, fn in (("ours", ours), ("hipblaslt", lib)):
                ms, clk, w = window(fn, args.seconds)
                rec[impl].append(dict(ms=ms, tflops=fl / ms / 1e9, sclk_mhz=clk, watts=w))
                print(f"{name:26s} {form} M={M} N={N} K={K} round {r} {impl:10s}: {ms:.3f} ms {fl / ms / 1e9:7.0f} TF/s  sclk {clk} MHz  {w} W",
                      flush=True)
        med = {k: sorted(v, key=lambda x: x["ms"])[len(v) // 2] for k, v in rec.items()}
        summary[name] = dict(form=form, M=M, N=N, K=K, ours=med["ours"], hipblaslt=med["hipblaslt"],
                             ours_over_hipblaslt=med["hipblaslt"]["ms"] / med["ours"]["ms"])
        torch.cuda.empty_cache()
    print("SUMMARY " + json.dumps(summary))


if __name__ == "__main__":
    main()
