"""NT 256x256 ping-pong kernel with parts removed (experiment library, results wrong by construction) + power / clock
samples: which part of the loop costs throughput under the 1.4 kW cap?  variant 3 = full, 101 = no LDS-DMA, 102 = no LDS
fragment reads, 103 = DMA always re-reads K tile 0 (every load an L2 hit).
Usage: RV_HIP_LIB=rlaif-v_amd/librlaifv_hip_exp.so python tools/exp_gemm_ablate_power.py"""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from rlaif_v_amd import ops  # noqa: E402

BF = torch.bfloat16


def sample(stop, rows):
    while not stop.is_set():
        try:
            d = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True,
                                          timeout=10).stdout).get("card0", {})
            rows.append((d.get("sclk clock speed:"), d.get("Current Socket Graphics Package Power (W)")))
        except Exception as e:      # noqa: BLE001
            rows.append(("err", repr(e)[:60]))
        time.sleep(0.05)


R, N, K = 27664, 22016, 4096
x = torch.randn(R, K, device="cuda").to(BF)
w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
out = torch.empty(R, N, device="cuda", dtype=BF)
for v, name in ((3, "full"), (101, "no LDS-DMA"), (102, "no LDS fragment reads"), (103, "DMA L2-resident")):
    rows, stop = [], threading.Event()
    th = threading.Thread(target=sample, args=(stop, rows))
    for _ in range(3):
        ops.gemm_nt(x, w, out=out, variant=v)
    torch.cuda.synchronize()
    t0 = time.time()
    th.start()
    n = 0
    while time.time() - t0 < 3.0:
        for _ in range(20):
            ops.gemm_nt(x, w, out=out, variant=v)
        torch.cuda.synchronize()
        n += 20
    dt = time.time() - t0
    stop.set()
    th.join()
    print(f"variant {v:3d} {name:24s}: {2.0 * R * N * K * n / dt / 1e12:7.0f} TF/s; sclk / W samples: {rows[2:8]}", flush=True)
