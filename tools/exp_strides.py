"""Stride experiments: does padding row strides away from powers of two help GEMM / attention?"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import ops, hip
BF = torch.bfloat16
dev = torch.device("cuda:0")


def timeit(fn, iters=8, warmup=2):
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


def view(rows, cols, pad):
    return torch.randn(rows, cols + pad, device=dev).to(BF)[:, :cols]


print("== GEMM, interleaved rounds (median of 3 rounds)")
for (M, N, K) in [(16384, 4096, 4096), (16384, 22016, 4096), (16384, 4096, 11008), (12288, 4096, 16384)]:
    res = {}
    bufs = {p: (view(M, K, p), view(N, K, p), torch.empty(M, N + p, dtype=BF, device=dev)[:, :N]) for p in (0, 64, 128, 192)}
    for rnd in range(3):
        for p, (a, b, c) in bufs.items():
            res.setdefault(p, []).append(timeit(lambda: ops.gemm_nt(a, b, out=c)))
    print(f"{M}x{N}x{K}: " + "  ".join(f"pad{p}: {sorted(v)[1]:.3f}ms {2.0*M*N*K/sorted(v)[1]/1e9:.0f}TF" for p, v in res.items()), flush=True)
    del bufs

print("== attention S=8 L=2048 H=32, qkv/do/o row padding")
S, L, H, hd = 8, 2048, 32, 128
d = H * hd
for pad in (0, 64, 128):
    qkv = view(S * L, 3 * d, pad)
    do = view(S * L, d, pad)
    o = torch.empty(S * L, d + pad, dtype=BF, device=dev)[:, :d]
    _, lse = ops.attn_fwd(qkv, S, L, H, hd, True, 0, d, 2 * d, out=o)
    t_f = timeit(lambda: ops.attn_fwd(qkv, S, L, H, hd, True, 0, d, 2 * d, out=o))
    delta = torch.empty(S, H, L, dtype=torch.float32, device=dev)
    hip.call("rv_attn_delta", do, do.stride(0), o, o.stride(0), delta, S, L, H, hd)
    dqkv = torch.empty(S * L, 3 * d + pad, dtype=BF, device=dev)[:, :3 * d]
    t_b = timeit(lambda: hip.call("rv_attn_bwd", qkv, qkv.stride(0), 0, d, 2 * d, do, do.stride(0), lse,
                                  delta, dqkv, dqkv.stride(0), S, L, H, hd, 1, 1.0 / math.sqrt(hd), None, None))
    print(f"pad {pad}: fwd {t_f:.3f} ms  bwd {t_b:.3f} ms", flush=True)
