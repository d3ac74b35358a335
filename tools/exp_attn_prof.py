"""Phase profile of the attention FORWARD and dQ kernels from s_memtime stamps (experiment build -DRV_ATTN_PROF): shader clocks per
64-key tile and phase at the bench shape, wave 0 of every workgroup (two workgroups share a CU, two waves a SIMD: a phase's ticks
include what the partner wave executed meanwhile).

    python tools/exp_attn_prof.py --build      (here: rlaif-v_amd/librlaifv_hip_aprof.so)
    RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_aprof.so python tools/exp_attn_prof.py     (GPU box)"""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if "--build" in sys.argv:
    import importlib.util
    spec = importlib.util.spec_from_file_location("b", os.path.join(REPO, "rlaif-v_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    os.environ["RV_BUILD_ONLY"] = "attention.hip"
    b.build_extension()
    print(b.build_extension(force=True, verbose=False, defines=("RV_ATTN_PROF",), tag="_aprof"))
    sys.exit(0)
from rlaif_v_amd import ops, hip  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda:0")
B, H, hd, shared, tail = 8, 32, 128, 638, 1410
L, d = shared + 2 * tail, H * hd
qkv = (torch.randn(B * L, 3 * d, device=dev) * 0.5).to(BF)
do = (torch.randn(B * L, d, device=dev) * 0.5).to(BF)
seg = (torch.full((B,), shared, dtype=torch.int32, device=dev), torch.full((B,), shared + tail, dtype=torch.int32, device=dev))
o, lse = ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, seg=seg)
dqkv = torch.empty_like(qkv)
lib = hip.lib().lib
lib.rv_debug_attn_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 16)()


def run(iters):
    for _ in range(iters):
        ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, out=o, seg=seg)
        ops.attn_bwd(qkv, o, do, lse, B, L, H, hd, True, 0, d, 2 * d, dqkv=dqkv, seg=seg)
    torch.cuda.synchronize()


run(3)
lib.rv_debug_attn_prof(buf)
iters = 10
run(iters)
lib.rv_debug_attn_prof(buf)
v = [int(x) for x in buf]
# tiles a query block visits (forward / dQ): key tiles up to its diagonal, minus the chosen-branch tiles of a rejected-branch block
e1 = shared + tail
nqb = (L + 127) // 128
tiles = 0
for qb in range(nqb):
    q0 = qb * 128
    nt = (min(L, q0 + 128) + 63) // 64
    skip = max((e1 >> 6) - ((shared + 63) >> 6), 0) if (q0 >= e1 and e1 > shared) else 0
    tiles += nt - skip
tiles_total = tiles * B * H
names = ["DMA issue (next tile)", "S^T [+ dP^T] MFMAs", "softmax / dS + second MFMA phase", "wait: next tile landed (vmcnt 0)", "barrier",
         "pass prologue / epilogue"]
for base, kname in ((0, "attn_fwd2_kernel"), (8, "attn_bwd_dq2_kernel")):
    x = v[base:base + 8]
    tot = sum(x[:6])
    print(f"{kname}: {x[7] / iters:.0f} workgroups, {tiles_total} tiles per launch, {x[6] / max(x[7], 1):.0f} ticks per workgroup (stamped {tot / max(x[7], 1):.0f})")
    for n, t in zip(names, x[:6]):
        print(f"  {n:36s} {100.0 * t / max(tot, 1):5.1f} %   {t / iters / tiles_total:8.2f} ticks per tile")
    print(f"  total per tile {tot / iters / tiles_total:.2f} ticks")
