"""Do the attention kernels camp on a few L2 channels?  Their operands are head slices (256 B per row) of token-major
buffers whose row strides are 24 KB (qkv, dqkv) and 8 KB (o, dO): every row of a head starts at the SAME address modulo
4 / 8 / 24 KB.  The bench shape (8 packed pair rows of 3458 tokens, 32 heads) is timed with the row strides padded by
`pad` elements (pad = 0: the layout of the training step).  Usage: python tools/exp_attn_strides.py [pads ...]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda:0")


def timeit(fn, iters=10, warmup=2):
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


def view(rows, cols, pad, scale=0.5):
    return (torch.randn(rows, cols + pad, device=dev) * scale).to(BF)[:, :cols]


def main():
    pads = [int(x) for x in sys.argv[1:]] or [0, 64, 128, 192, 0, 64]
    B, H, hd, shared, tail = 8, 32, 128, 638, 1410
    L, d = shared + 2 * tail, H * hd
    seg = (torch.full((B,), shared, dtype=torch.int32, device=dev), torch.full((B,), shared + tail, dtype=torch.int32, device=dev))
    for _ in range(2):          # clock ramp: the first measurements of a process run slower
        a = view(4096, 4096, 0)
        timeit(lambda: ops.gemm_nt(a, a), 5)
    for pad in pads:
        qkv, do = view(B * L, 3 * d, pad), view(B * L, d, pad)
        o = torch.empty(B * L, d + pad, dtype=BF, device=dev)[:, :d]
        dqkv = torch.empty(B * L, 3 * d + pad, dtype=BF, device=dev)[:, :3 * d]
        _, lse = ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, out=o, seg=seg)
        t_f = timeit(lambda: ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, out=o, seg=seg))
        t_b = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, B, L, H, hd, True, 0, d, 2 * d, dqkv=dqkv, seg=seg))
        print(f"pad {pad:4d} (row strides {2 * (3 * d + pad)} / {2 * (d + pad)} B): fwd {t_f:.3f} ms  bwd (dq + dkv) {t_b:.3f} ms", flush=True)
        del qkv, do, o, dqkv


if __name__ == "__main__":
    main()
