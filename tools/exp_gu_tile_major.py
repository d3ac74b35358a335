#!/usr/bin/env python
"""the kept gate|up tensor of the fused SwiGLU epilogues TILE-MAJOR (256 x 256 tiles contiguous, RV_GU_TILE_MAJOR=1)
against row-major, at the step's shape (27,664 x 22,016 x 4,096 forward, 27,664 x 11,008 x 4,096 backward).  Each layout in its own
process (the switch is read once), two processes each, three timing rounds; the SHA-256 of act and d(gate|up) shows the two layouts
compute the same thing.  The switch exists in the EXPERIMENT library only:
    python rlaif-v_amd/build.py --experiments ; RV_HIP_LIB=$PWD/rlaif-v_amd/librlaifv_hip_exp.so python tools/exp_gu_tile_major.py   (GPU box)"""
import hashlib
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from rlaif_v_amd import ops
    from tools.bench_hot_kernels import timeit
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    R, d, f = 27664, 4096, 11008
    x = (torch.randn(R, d, device=dev) * 0.5).to(torch.bfloat16)
    wguT = (torch.randn(d, 2 * f, device=dev) * 0.02).to(torch.bfloat16)
    dy = (torch.randn(R, d, device=dev) * 0.5).to(torch.bfloat16)
    wdown = (torch.randn(d, f, device=dev) * 0.02).to(torch.bfloat16)
    gu, act = ops.linear_swiglu(x, wguT)
    dgu = ops.linear_swiglu_bwd(dy, wdown, gu)
    torch.cuda.synchronize()
    sha = lambda t: hashlib.sha256(t.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
    s_act, s_dgu = sha(act), sha(dgu)
    fw = [timeit(lambda: ops.linear_swiglu(x, wguT), 10) for _ in range(3)]
    bw = [timeit(lambda: ops.linear_swiglu_bwd(dy, wdown, gu), 10) for _ in range(3)]
    print(f"tile_major={os.environ.get('RV_GU_TILE_MAJOR', '0')}: swiglu fwd {' '.join(f'{m:.3f}' for m in fw)} ms | swiglu bwd "
          f"{' '.join(f'{m:.3f}' for m in bw)} ms ({2.0 * R * f * d / bw[-1] / 1e9:.0f} TF/s) | act {s_act} dgu {s_dgu}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for rnd in range(2):
            for tm in ("0", "1"):
                subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, RV_GU_TILE_MAJOR=tm))
