import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import ops
BF = torch.bfloat16; dev = torch.device("cuda:0")
def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (M, N, K) in [(16384, 4096, 4096), (8192, 8192, 8192), (16384, 22016, 4096)]:
    a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF)
    z = torch.zeros(M, K, dtype=BF, device=dev); zb = torch.zeros(N, K, dtype=BF, device=dev)
    c = torch.empty(M, N, dtype=BF, device=dev)
    ref = ops.gemm_nt(a, b, variant=1)
    res = {}
    for rnd in range(3):
        for name, v, aa, bb in [("v3", 3, a, b), ("v6 split", 6, a, b), ("v2 Lseg", 2, a, b), ("no-DMA", 101, a, b)]:
            res.setdefault(name, []).append(timeit(lambda: ops.gemm_nt(aa, bb, out=c, variant=v)))
            if v in (4, 5, 6) and rnd == 0:
                print(f"v{v} exact:", torch.equal(c, ref))
    for name, v in res.items():
        ms = sorted(v)[1]
        print(f"{M}x{N}x{K} {name:14s}: {ms:.3f} ms {2.0*M*N*K/ms/1e9:.0f} TF/s", flush=True)
