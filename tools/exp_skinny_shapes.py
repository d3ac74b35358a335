#!/usr/bin/env python
"""Round-6 measurement: the rank-64 skinny products of config 5's step, shape by shape, against the HBM time of their operands
(M = 30,216 rows).  NT: t = x A^T / dt = dy B (gemm_nt -> gemm_nt_skinny_kernel); TN: dA = dt^T xd / dB = dy^T t (gemm_tn_skinny)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import ops  # noqa: E402

dev = "cuda:0"
M = int(os.environ.get("M", 30216))
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16, generator=g) * 0.05


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = [0.0, 0.0]
print(f"{'product':44s} {'us':>8s} {'GB':>7s} {'TB/s':>6s}")
for name, K, N, cnt in [("t = x A^T      K=4096  N=192 (qkv)", 4096, 192, 1), ("t = x A^T      K=4096  N=64  (o)", 4096, 64, 1),
                        ("t = x A^T      K=4096  N=128 (gate|up)", 4096, 128, 1), ("t = act A^T    K=11008 N=64  (down)", 11008, 64, 1),
                        ("dt = dy B      K=4096  N=64  (q / k / v / o / down, x5)", 4096, 64, 5),
                        ("dt = dy B      K=11008 N=64  (gate / up, x2)", 11008, 64, 2)]:
    a, b = rn(M, K), rn(N, K)
    us = timeit(lambda: ops.gemm_nt(a, b, alpha=0.25))
    gb = 2.0 * (a.numel() + b.numel() + M * N) / 1e9
    print(f"NT {name:41s} {us:8.1f} {gb:7.3f} {gb / us * 1e3:6.2f}")
    tot[0] += us * cnt
    tot[1] += gb * cnt
    del a, b
for name, I, J, cnt in [("dA = dt^T xd   [192 x 4096] (qkv)", 192, 4096, 1), ("dA = dt^T xd   [64 x 4096] (o)", 64, 4096, 1),
                        ("dA = dt^T xd   [128 x 4096] (gate|up)", 128, 4096, 1), ("dA = dt^T xd   [64 x 11008] (down)", 64, 11008, 1),
                        ("dB = dy^T t    [4096 x 64] (q / k / v / o / down, x5)", 4096, 64, 5),
                        ("dB = dy^T t    [11008 x 64] (gate / up, x2)", 11008, 64, 2)]:
    p, q = rn(M, I), rn(M, J)
    out = torch.empty(I, J, dtype=torch.bfloat16, device=dev)
    us = timeit(lambda: ops.gemm_tn_skinny(p, q, out=out))
    gb = 2.0 * (p.numel() + q.numel() + I * J) / 1e9
    print(f"TN {name:41s} {us:8.1f} {gb:7.3f} {gb / us * 1e3:6.2f}")
    tot[0] += us * cnt
    tot[1] += gb * cnt
    del p, q
print(f"per layer: {tot[0]:.0f} us for {tot[1]:.2f} GB = {tot[1] / tot[0] * 1e3:.2f} TB/s; x 32 layers = {tot[0] * 32 / 1e3:.1f} ms per step")
