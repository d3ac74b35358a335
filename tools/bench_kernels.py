"""Kernel micro-benchmarks on one MI355X (random data, HIP-event timing on the launch stream).
Usage: python tools/bench_kernels.py [--quick]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--tokens", type=int, default=16384)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = []
    N = args.tokens
    shapes = [("qkv", N, 12288, 4096), ("o", N, 4096, 4096), ("gate_up", N, 22016, 4096), ("down", N, 4096, 11008),
              ("wgrad_qkv", 12288, 4096, N), ("wgrad_down", 4096, 11008, N), ("lm_head", 8192, 32000, 4096),
              ("square4k", 4096, 4096, 4096), ("square8k", 8192, 8192, 8192)]
    if args.quick:
        shapes = shapes[:2] + shapes[-2:-1]
    for name, M, Nn, K in shapes:
        a = torch.randn(M, K, device=dev).to(BF)
        b = torch.randn(Nn, K, device=dev).to(BF)
        out = torch.empty(M, Nn, dtype=BF, device=dev)
        for variant in (1, 2, 3):
            ms = timeit(lambda: ops.gemm_nt(a, b, out=out, variant=variant))
            tf = 2.0 * M * Nn * K / ms / 1e9
            res.append(dict(kernel="gemm_nt", name=name, variant=variant, M=M, N=Nn, K=K, ms=ms, tflops=tf))
            print(f"gemm {name:12s} v{variant} {M}x{Nn}x{K}: {ms:8.3f} ms  {tf:8.1f} TF/s", flush=True)
        del a, b, out
    for name, R, I, J in [("wgrad_qkv", N, 12288, 4096), ("wgrad_gu", N, 22016, 4096), ("wgrad_down", N, 4096, 11008)]:
        pp = torch.randn(R, I, device=dev).to(BF)
        qq = torch.randn(R, J, device=dev).to(BF)
        out = torch.empty(I, J, dtype=BF, device=dev)
        ms = timeit(lambda: ops.gemm_tn(pp, qq, out=out))
        ms2 = timeit(lambda: ops.gemm_nt(ops.transpose(pp), ops.transpose(qq), out=out))
        print(f"gemm_tn {name:11s} {R}x{I}x{J}: {ms:8.3f} ms {2.0 * R * I * J / ms / 1e9:8.1f} TF/s   (transpose x2 + NT: {ms2:.3f} ms)", flush=True)
        res.append(dict(kernel="gemm_tn", name=name, ms=ms, tflops=2.0 * R * I * J / ms / 1e9, ms_transpose_nt=ms2))
        del pp, qq, out
    # attention, LLaVA shape
    S, L, H, hd = (8, 2048, 32, 128)
    qkv = torch.randn(S * L, 3 * H * hd, device=dev).to(BF)
    do = torch.randn(S * L, H * hd, device=dev).to(BF)
    out = torch.empty(S * L, H * hd, dtype=BF, device=dev)
    ms = timeit(lambda: ops.attn_fwd(qkv, S, L, H, hd, True, 0, H * hd, 2 * H * hd, out=out))
    fl = 4.0 * S * H * L * L * hd / 2
    print(f"attn_fwd causal S{S} L{L}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF/s (causal-halved flops)", flush=True)
    res.append(dict(kernel="attn_fwd", ms=ms, tflops=fl / ms / 1e9))
    o, lse = ops.attn_fwd(qkv, S, L, H, hd, True, 0, H * hd, 2 * H * hd)
    dqkv = torch.empty_like(qkv)
    ms = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, S, L, H, hd, True, 0, H * hd, 2 * H * hd, dqkv=dqkv), iters=5)
    print(f"attn_bwd (delta+3 transposes+dq+dkv): {ms:8.3f} ms  {2.5 * fl / ms / 1e9:8.1f} TF/s (2.5x fwd flops)", flush=True)
    res.append(dict(kernel="attn_bwd", ms=ms, tflops=2.5 * fl / ms / 1e9))
    # HBM-bound kernels
    x = torch.randn(N, 4096, device=dev).to(BF)
    w = torch.ones(4096, dtype=BF, device=dev)
    y = torch.empty_like(x)
    ms = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-5, out=y))
    print(f"rmsnorm_fwd {N}x4096: {ms:8.3f} ms  {2 * x.numel() * 2 / ms / 1e6:8.1f} GB/s", flush=True)
    res.append(dict(kernel="rmsnorm_fwd", ms=ms, gbs=2 * x.numel() * 2 / ms / 1e6))
    t = torch.empty(4096, ops.round_up(N, 64), dtype=BF, device=dev)
    ms = timeit(lambda: ops.transpose(x, out=t))
    print(f"transpose {N}x4096: {ms:8.3f} ms  {2 * x.numel() * 2 / ms / 1e6:8.1f} GB/s", flush=True)
    res.append(dict(kernel="transpose", ms=ms, gbs=2 * x.numel() * 2 / ms / 1e6))
    gu = torch.randn(N, 22016, device=dev).to(BF)
    act = torch.empty(N, 11008, dtype=BF, device=dev)
    ms = timeit(lambda: ops.swiglu_fwd(gu, out=act))
    print(f"swiglu_fwd: {ms:8.3f} ms  {(gu.numel() + act.numel()) * 2 / ms / 1e6:8.1f} GB/s", flush=True)
    n = 8 * 1024 * 1024 * 16
    p = torch.zeros(n, dtype=BF, device=dev)
    m32, mm, vv = (torch.zeros(n, device=dev) for _ in range(3))
    g = torch.randn(n, device=dev).to(BF)
    ms = timeit(lambda: ops.adamw_step(p, m32, mm, vv, g, 1e-6, 0.9, 0.999, 1e-8, 0.01, 1))
    print(f"adamw {n / 1e6:.0f}M params: {ms:8.3f} ms  {n * 28 / ms / 1e6:8.1f} GB/s", flush=True)
    res.append(dict(kernel="adamw", ms=ms, gbs=n * 28 / ms / 1e6))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bench_kernels.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
