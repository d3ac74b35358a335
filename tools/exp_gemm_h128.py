"""the NN GEMMs as two 4-wave workgroups per CU on 128 x 256 tiles (gemm_nn_h128_kernel, -DRV_GEMM_H128 builds).

    python tools/exp_gemm_h128.py --build        (here: rlaif-v_amd/librlaifv_hip_h128.so)
    bash tools/calls/r04_call21.sh               (GPU box: A/B through tools/exp_gemm_lib_ab.py, RV_H128 = 0 | 1, staggers)
    RV_HIP_LIB=... RV_H128=1 python tools/exp_gemm_h128.py --ldsalloc      (dump HW_REG_LDS_ALLOC of the first workgroups)"""
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
    b.build_extension()
    os.environ["RV_BUILD_ONLY"] = "gemm.hip"
    print(b.build_extension(force=True, verbose=False, defines=("RV_GEMM_H128",), tag="_h128"))
    sys.exit(0)

import torch  # noqa: E402
from rlaif_v_amd import hip, ops  # noqa: E402

if "--ldsalloc" in sys.argv:
    dev = torch.device("cuda:0")
    lib = hip.lib().lib
    M, N, K = 27664, 4096, 4096
    tiles = ((M + 127) // 128) * (N // 256)
    dbg = torch.zeros(tiles, dtype=torch.int32, device=dev)
    lib.rv_debug_h128_dbg.argtypes = [ctypes.c_void_p]
    lib.rv_debug_h128_dbg(ctypes.c_void_p(dbg.data_ptr()))
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    ops.gemm_nn(a, b)
    torch.cuda.synchronize()
    lib.rv_debug_h128_dbg(ctypes.c_void_p(0))
    v = dbg.cpu().tolist()
    print("HW_REG_LDS_ALLOC of workgroups 0..15:   ", [hex(x & 0xffffffff) for x in v[:16]])
    print("HW_REG_LDS_ALLOC of workgroups 256..271:", [hex(x & 0xffffffff) for x in v[256:272]])
    print("HW_REG_LDS_ALLOC of workgroups 512..527:", [hex(x & 0xffffffff) for x in v[512:528]])
    first = v[:512]
    nz = sum(1 for x in first if (x & 0xfff) != 0)
    print(f"first 512 workgroups: {nz} with a non-zero LDS base; distinct values {sorted(set(hex(x & 0xffffffff) for x in v))[:8]}")
