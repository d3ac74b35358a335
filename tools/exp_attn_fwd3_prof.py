"""Phase profile of attn_fwd3_kernel from s_memtime stamps (experiment build -DRV_ATTN_PROF), bench shape (8 packed rows x 3,458):
shader clocks per 64-key tile and phase, wave 0 of every workgroup (ONE wave per SIMD: a phase's ticks are the wave's own).
    python tools/exp_attn_fwd3_prof.py --build ;  RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_aprof.so python tools/exp_attn_fwd3_prof.py"""
import ctypes
import os
import sys

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
import torch  # noqa: E402
from rlaif_v_amd import hip, ops  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda:0")
B, H, hd, shared, tail = 8, 32, 128, 638, 1410
L, d = shared + 2 * tail, H * hd
qkv = (torch.randn(B * L, 3 * d, device=dev) * 0.5).to(BF)
seg = (torch.full((B,), shared, dtype=torch.int32, device=dev), torch.full((B,), shared + tail, dtype=torch.int32, device=dev))
hip.lib().call("rv_set_attn_fwd_version", 3)
o, lse = ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, seg=seg)
lib = hip.lib().lib
lib.rv_debug_attn_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, out=o, seg=seg)
torch.cuda.synchronize()
lib.rv_debug_attn_prof(buf)
iters = 10
for _ in range(iters):
    ops.attn_fwd(qkv, B, L, H, hd, True, 0, d, 2 * d, out=o, seg=seg)
torch.cuda.synchronize()
lib.rv_debug_attn_prof(buf)
v = [int(x) for x in buf]
e1 = shared + tail
tiles = 0
for qb in range((L + 255) // 256):
    q0 = qb * 256
    nt = (min(L, q0 + 256) + 63) // 64
    skip = max((e1 >> 6) - ((shared + 63) >> 6), 0) if (q0 >= e1 and e1 > shared) else 0
    tiles += nt - skip
tiles *= B * H
wg = v[7] / iters
names = ["phase 1  PV_B || softmax_A first", "per-tile set-up (dispatch, DMA addresses, masks)", "wait + barrier + phase 2  S_B || softmax_A rest + K DMA", "phase 3  PV_A || softmax_B first",
         "phase 4  S_A(+K reads) || softmax_B rest + V DMA", "pass prologue / epilogue"]
tot = sum(v[:6])
print(f"attn_fwd3_kernel: {wg:.0f} workgroups, {tiles} tiles (256 q x 64 k) per launch, {v[6] / v[7]:.0f} ticks per workgroup")
for n, x in zip(names, v[:6]):
    print(f"  {n:50s} {100.0 * x / tot:5.1f} %   {x / iters / tiles:8.1f} ticks per tile")
print(f"  total per tile {tot / iters / tiles:.1f} ticks (16 MFMAs per phase = 512 matrix-pipe clocks, 2048 per tile)")
