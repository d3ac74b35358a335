"""One library (RV_HIP_LIB) on the step's four NN GEMM shapes: time per launch + a bit-level checksum of every output, so
experiment builds (epilogue prefetch, sc1 / nt stores, persistent tile loop) can be compared with the shipped library for
speed AND for identical results.  Usage: RV_HIP_LIB=... python tools/exp_gemm_lib_ab.py [--iters 8]"""
import argparse
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import hip, ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def digest(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.contiguous().view(torch.int16).cpu().numpy().tobytes())
    return h.hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--rows", type=int, default=27664)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    R, d, f = a.rows, 4096, 11008
    g = torch.Generator(device=dev).manual_seed(0)

    def rn(*shape, s=1.0):
        return (torch.randn(*shape, device=dev, generator=g) * s).to(BF)
    x, wguT, wdown, wdownT, wqkvT = rn(R, d), rn(d, 2 * f, s=0.02), rn(d, f, s=0.02), rn(f, d, s=0.02), rn(d, 3 * d, s=0.02)
    gu, act = rn(R, 2 * f), rn(R, f)
    out_qkv, out_d = torch.empty(R, 3 * d, device=dev, dtype=BF), torch.empty(R, d, device=dev, dtype=BF)
    print(f"library: {hip.lib().path}", flush=True)
    res = {}
    res["swiglu"] = ops.linear_swiglu(x, wguT)
    res["swiglu_bwd"] = (ops.linear_swiglu_bwd(x, wdown, gu),)
    res["qkv"] = (ops.gemm_nn(x, wqkvT, out=out_qkv).clone(),)
    res["down+res"] = (ops.gemm_nn(act, wdownT, out=out_d, residual=x).clone(),)
    torch.cuda.synchronize()
    cases = [("swiglu", lambda: ops.linear_swiglu(x, wguT), 2.0 * R * 2 * f * d),
             ("swiglu_bwd", lambda: ops.linear_swiglu_bwd(x, wdown, gu), 2.0 * R * f * d),
             ("qkv", lambda: ops.gemm_nn(x, wqkvT, out=out_qkv), 2.0 * R * 3 * d * d),
             ("down+res", lambda: ops.gemm_nn(act, wdownT, out=out_d, residual=x), 2.0 * R * d * f)]
    for rep in range(2):                  # two rounds: the second one is the number to read (clocks settled)
        row = []
        for name, fn, fl in cases:
            ms = timeit(fn, a.iters)
            row.append(f"{name} {ms:6.3f} ms {fl / ms / 1e9:6.0f} TF/s")
        print(f"round {rep}: " + " | ".join(row), flush=True)
    print("checksums: " + " ".join(f"{k}={digest(*v)}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
