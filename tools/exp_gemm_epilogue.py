"""Epilogue / scheduling experiments on the 64-deep-A NN GEMM at the step's shapes:
XCD stagger (rv_set_gemm_tuning bits 0-7), serpentine column order (bit 16), and - by running the script once per library
(RV_HIP_LIB=rlaif-v_amd/librlaifv_hip_sc1.so / _nt.so, built with -DRV_EPI_CPOL=1|2) - the cache policy of the output stores.
Usage: python tools/exp_gemm_epilogue.py [--iters 6]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import hip, ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters, warmup=2):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--staggers", default="0,1,2,3,4,6")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    R, d, f = 27664, 4096, 11008
    x = torch.randn(R, d, device=dev).to(BF)
    wguT = (torch.randn(d, 2 * f, device=dev) * 0.02).to(BF)
    wdown = (torch.randn(d, f, device=dev) * 0.02).to(BF)          # [out = d][in = f]: the dgrad GEMM contracts over d
    wdownT = (torch.randn(f, d, device=dev) * 0.02).to(BF)
    wqkvT = (torch.randn(d, 3 * d, device=dev) * 0.02).to(BF)
    gu = torch.randn(R, 2 * f, device=dev).to(BF)
    act = torch.randn(R, f, device=dev).to(BF)
    out_qkv = torch.empty(R, 3 * d, device=dev, dtype=BF)
    out_d = torch.empty(R, d, device=dev, dtype=BF)
    cases = [
        ("nn_swiglu     gate|up  K=4096 ", lambda: ops.linear_swiglu(x, wguT), 2.0 * R * 2 * f * d),
        ("nn_swiglu_bwd d act    K=4096 ", lambda: ops.linear_swiglu_bwd(x, wdown, gu), 2.0 * R * f * d),
        ("nn plain      qkv      K=4096 ", lambda: ops.gemm_nn(x, wqkvT, out=out_qkv), 2.0 * R * 3 * d * d),
        ("nn +residual  down     K=11008", lambda: ops.gemm_nn(act, wdownT, out=out_d, residual=x), 2.0 * R * d * f),
    ]
    lib = hip.lib()
    print(f"library: {lib.path}", flush=True)
    for serp in (0, 1):
        for stg in [int(v) for v in a.staggers.split(",")]:
            lib.call("rv_set_gemm_tuning", 0, stg | (serp << 16))
            row = []
            for name, fn, fl in cases:
                ms = timeit(fn, a.iters)
                row.append(f"{name.split()[0]} {ms:6.3f} ms {fl / ms / 1e9:6.0f} TF/s")
            print(f"serp {serp} stagger {stg}: " + " | ".join(row), flush=True)
    lib.call("rv_set_gemm_tuning", 0, 0)


if __name__ == "__main__":
    main()
