"""A/B of the LDS ring depth after the vmcnt fix: NT variant 3 (DIST 3) vs 4 (DIST 4); TN via RV_GEMM_TN_DIST."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlaif_v_amd import ops
BF = torch.bfloat16
def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
R = 27664
mode = sys.argv[1] if len(sys.argv) > 1 else "nt"
if mode == "nt":
    for name, M, N, K in [("qkv", R, 12288, 4096), ("o", R, 4096, 4096), ("gate_up", R, 22016, 4096), ("down", R, 4096, 11008)]:
        a = torch.randn(M, K, device="cuda").to(BF); b = torch.randn(N, K, device="cuda").to(BF); out = torch.empty(M, N, device="cuda", dtype=BF)
        bT = b.t().contiguous()
        for rep in range(2):
            ms = timeit(lambda: ops.gemm_nt(a, b, out=out, variant=3))
            print(f"nt {name:8s} v3: {ms:.3f} ms {2*M*N*K/ms/1e9:7.1f} TF/s")
            ms = timeit(lambda: ops.gemm_nn(a, bT, out=out))
            print(f"nn {name:8s}   : {ms:.3f} ms {2*M*N*K/ms/1e9:7.1f} TF/s")
else:
    for name, I, J in [("wqkv", 12288, 4096), ("wdown", 4096, 11008), ("wgu", 22016, 4096)]:
        p = torch.randn(R, I, device="cuda").to(BF); q = torch.randn(R, J, device="cuda").to(BF); out = torch.empty(I, J, device="cuda", dtype=BF)
        for rep in range(2):
            ms = timeit(lambda: ops.gemm_tn(p, q, out=out))
            print(f"tn dist={os.environ.get('RV_GEMM_TN_DIST','3')} {name:6s}: {ms:.3f} ms {2*R*I*J/ms/1e9:7.1f} TF/s")
