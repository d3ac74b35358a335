"""Forward and backward attention time at the bench step's packed shape (8 rows: 638 shared + 2 x 1410 tokens), for A/B runs of
experiment libraries (RV_HIP_LIB).  Usage: [RV_HIP_LIB=...] python tools/exp_attn_packed_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import hip, ops  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda:0")
B, H, hd, shared, tail = 8, 32, 128, 638, 1410
L, d = shared + 2 * tail, H * hd
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B * L, 3 * d, device=dev, generator=g) * 0.5).to(BF)
do = (torch.randn(B * L, d, device=dev, generator=g) * 0.5).to(BF)
seg = (torch.full((B,), shared, dtype=torch.int32, device=dev), torch.full((B,), shared + tail, dtype=torch.int32, device=dev))
o, lse = ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, seg=seg)
dqkv = torch.empty_like(qkv)


def timeit(fn, iters=20, warmup=5):
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


print(f"library: {hip.lib().path}")
for rep in range(3):
    tf = timeit(lambda: ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, out=o, seg=seg))
    tb = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, B, L, H, hd, True, 0, d, 2 * d, dqkv=dqkv, seg=seg))
    print(f"  forward {tf:.4f} ms   backward (dQ + dK/dV) {tb:.4f} ms", flush=True)
print("checksum dqkv:", float(dqkv.float().abs().sum()))
