#!/usr/bin/env python
"""Round-6 experiment: what the SwiGLU epilogue of the fused-LoRA gate|up GEMM costs on config 5's shape (M = 30,216 rows,
N = 2 x 11008, K = 4096, adapter K2 = 128), against the plain fused-LoRA GEMM + the stand-alone SwiGLU(+dropout) kernel."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import ops  # noqa: E402

dev = "cuda:0"
M, f, d, rp = int(os.environ.get("M", 30216)), 11008, 4096, 64
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, d, device=dev, dtype=torch.bfloat16, generator=g)
wT = (torch.randn(d, 2 * f, device=dev, dtype=torch.bfloat16, generator=g) * 0.02)
w = wT.t().contiguous()
t2 = torch.randn(M, 2 * rp, device=dev, dtype=torch.bfloat16, generator=g) * 0.1
bexp = torch.randn(2 * rp, 2 * f, device=dev, dtype=torch.bfloat16, generator=g) * 0.02
B = bexp[:rp].t().contiguous()          # [2f, rp] stand-in adapter for the block-layout call
BT = bexp[:rp].contiguous()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


fl = 2.0 * M * 2 * f * d
r = {}
r["plain nn (no adapter)"] = timeit(lambda: ops.gemm_nn(x, wT))
r["fused-LoRA EXT, EpiStore (2 groups of K2 = 64)"] = timeit(lambda: ops.linear_lora(x, w, wT, t2, B, BT, group_cols=f))
gu = ops.linear_lora(x, w, wT, t2, B, BT, group_cols=f)
r["swiglu_fwd_dropout kernel"] = timeit(lambda: ops.swiglu_fwd_dropout(gu, 0.05, 7))
r["swiglu_fwd kernel"] = timeit(lambda: ops.swiglu_fwd(gu))
r["fused-LoRA EXT, EpiSwiGLU, p = 0 (gu + act)"] = timeit(lambda: ops.linear_lora_swiglu(x, wT, t2, bexp, 0.0, 0))
r["fused-LoRA EXT, EpiSwiGLU, p = 0.05 (gu + act + actd)"] = timeit(lambda: ops.linear_lora_swiglu(x, wT, t2, bexp, 0.05, 7))
r["full-FT EpiSwiGLU (gu + act)"] = timeit(lambda: ops.linear_swiglu(x, wT))
for k, v in r.items():
    print(f"{k:58s} {v:7.3f} ms   {fl / v / 1e9:7.1f} TF/s (base flops)")
