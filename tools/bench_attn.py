"""Attention kernel timing breakdown (fwd / dq+dkv / helpers) for several sequence lengths."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import ops, hip  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=5, warmup=2):
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
    dev = torch.device("cuda:0")
    S, H, hd = 8, 32, 128
    for L in [int(x) for x in (sys.argv[1:] or ["2048", "2112", "1984", "1024"])]:
        d = H * hd
        qkv = torch.randn(S * L, 3 * d, device=dev).to(BF)
        do = torch.randn(S * L, d, device=dev).to(BF)
        o, lse = ops.attn_fwd(qkv, S, L, H, hd, True, 0, d, 2 * d)
        out = torch.empty_like(o)
        t_f = timeit(lambda: ops.attn_fwd(qkv, S, L, H, hd, True, 0, d, 2 * d, out=out))
        dqkv = torch.empty_like(qkv)
        t_b = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, S, L, H, hd, True, 0, d, 2 * d, dqkv=dqkv))
        fl = 4.0 * S * H * L * L * hd / 2
        print(f"L={L}: fwd {t_f:.3f} ms ({fl / t_f / 1e9:.0f} TF/s)  bwd(dq+dkv) {t_b:.3f} ms ({2.5 * fl / t_b / 1e9:.0f} TF/s)"
              f"  per-L^2: fwd {t_f / L / L * 1e6:.3f} bwd {t_b / L / L * 1e6:.3f}", flush=True)


if __name__ == "__main__":
    main()


def gemm_ld_experiment():
    """Does a power-of-two row stride hurt the GEMM operand fetch?  Same shape, ld = K vs K + 64."""
    dev = torch.device("cuda:0")
    for (M, N, K) in [(16384, 4096, 4096), (16384, 12288, 4096), (4096, 11008, 16384)]:
        for pad in (0, 64):
            a = torch.randn(M, K + pad, device=dev).to(BF)[:, :K]
            b = torch.randn(N, K + pad, device=dev).to(BF)[:, :K]
            c = torch.empty(M, N + pad, dtype=BF, device=dev)[:, :N]
            ms = timeit(lambda: ops.gemm_nt(a, b, out=c), iters=10)
            print(f"gemm {M}x{N}x{K} ld pad {pad}: {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.0f} TF/s", flush=True)


if __name__ == "__main__" and os.environ.get("GEMM_LD"):
    gemm_ld_experiment()
