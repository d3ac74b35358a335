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
for (M, N, K) in [(27664, 4096, 4096), (27664, 12288, 4096), (27664, 22016, 4096), (27664, 4096, 11008)]:
    a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF)
    c = torch.empty(M, N, dtype=BF, device=dev)
    bT = b.t().contiguous()
    ms = sorted(timeit(lambda: ops.gemm_nn(a, bT, out=c)) for _ in range(3))[1]
    print(f"group={os.environ.get('RV_GEMM_GROUP','8')} {M}x{N}x{K}: {ms:.3f} ms {2.0*M*N*K/ms/1e9:.0f} TF/s", flush=True)
