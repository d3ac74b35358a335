"""Tail split of the weight-gradient GEMMs (rv_gemm_tn_bf16_ws) vs the plain launch on the step's shapes.
Usage: RV_TN_TAIL_SPLIT=0|1 [RV_TN_TAIL_PENALTY=x] python tools/exp_tn_tail.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import hip, ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=8, warmup=3):
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


dev = torch.device("cuda:0")
R = 27664
print("RV_TN_TAIL_SPLIT", os.environ.get("RV_TN_TAIL_SPLIT", "1"), "penalty", os.environ.get("RV_TN_TAIL_PENALTY", "default"), flush=True)
for rep in range(2):
    row = []
    for name, I, J in [("wqkv", 12288, 4096), ("wo", 4096, 4096), ("wgu", 22016, 4096), ("wdown", 4096, 11008)]:
        p = torch.randn(R, I, device=dev).to(BF)
        q = torch.randn(R, J, device=dev).to(BF)
        out = torch.empty(I, J, device=dev, dtype=BF)
        ms = timeit(lambda: ops.gemm_tn(p, q, out=out))
        need = hip.lib().lib.rv_gemm_tn_workspace_floats(R, I, J)
        row.append(f"{name} {ms:.3f} ms {2.0 * R * I * J / ms / 1e9:6.0f} TF/s (splits x tail {need // 65536})")
        del p, q, out
    print(f"round {rep}: " + " | ".join(row), flush=True)
