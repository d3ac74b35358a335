"""Phase profile of attn_bwd_dkv5_kernel from its s_memtime stamps: ticks (shader clocks) per 64-query tile and phase at the bench
shape, wave 0 of every workgroup.  Needs the experiment build -DRV_DKV5_PROF:

    python tools/exp_dkv5_prof.py --build        (here: writes rlaif-v_amd/librlaifv_hip_prof5.so)
    RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_prof5.so python tools/exp_dkv5_prof.py      (on the GPU box)

(The same stamps on version 4 of the kernel and its ablation bodies - profiles/r04_attn_dkv4_*.log - are what version 5's schedule
was derived from; that kernel and its generator live in history, commit 5e465f4.)"""
import ctypes
import math
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
    print(b.build_extension(force=True, verbose=False, defines=("RV_DKV5_PROF",), tag="_prof5"))
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
lib.rv_debug_dkv5_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    ops.attn_bwd(qkv, o, do, lse, B, L, H, hd, True, 0, d, 2 * d, dqkv=dqkv, seg=seg)
torch.cuda.synchronize()
lib.rv_debug_dkv5_prof(buf)
iters = 10
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    ops.attn_bwd(qkv, o, do, lse, B, L, H, hd, True, 0, d, 2 * d, dqkv=dqkv, seg=seg)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
lib.rv_debug_dkv5_prof(buf)
v = [int(x) for x in buf]
# tiles a workgroup processes: sum over key blocks of (nt - t_begin)
nkb = (L + 127) // 128
e1 = shared + tail
tiles = 0
for kb in range(nkb):
    kv0 = kb * 128
    nt = min((L + 63) // 64, (e1 + 63) // 64) if (kv0 >= shared and kv0 + 127 < e1) else (L + 63) // 64
    tiles += nt - kv0 // 64
tiles_total = tiles * B * H
wgs = v[15] / iters
names = ["glue", "I  S^T_A dP^T_A (+DMA)", "II S^T_B dP^T_B (+exp A)", "III dV_A dK_A (+exp)", "IV dV_B dK_B (+next tile)", "wait for tile t+1", "barrier", "pass-end wait + barrier",
         "loop-end vmcnt(0) + barrier", "next pass issue (DMA, K/V loads)", "read-out + stores", "kernel prologue"]
tot = sum(v[:12])
print(f"lib {os.environ.get('RV_HIP_LIB', 'default')}: dq + dkv {ms:.3f} ms; {wgs:.0f} workgroups, {tiles_total} tiles per launch, "
      f"{v[14] / v[15]:.0f} ticks per workgroup, stamped {tot / v[15]:.0f}")
for n, x in zip(names, v[:12]):
    print(f"  {n:38s} {100.0 * x / tot:5.1f} %   {x / iters / tiles_total:8.2f} ticks per tile")
print(f"  total per tile {tot / iters / tiles_total:.2f} ticks (wave 0 of every workgroup)")
