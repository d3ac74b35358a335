"""Experiment: fabric-side (L2 miss) read traffic of the 256x256 NN GEMM as a function of how many rounds of workgroups a launch
has - does an XCD's set of concurrently running tiles keep sharing its A / B panels through the 4 MiB L2 once the workgroups
of later rounds start at staggered times?

    worker:  python tools/exp_gemm_l2_reuse.py worker          (run under rocprofv3 --pmc FETCH_SIZE --kernel-trace)
    report:  python tools/exp_gemm_l2_reuse.py report <results.db>

The ideal per launch is one fabric read of every operand panel per XCD-resident tile cluster (4 row tiles x 8 column tiles):
(4 + 8) * 256 rows * K * 2 B per 32 tiles.
"""
import importlib
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SHAPES = [(4096, 4096, 4096), (8192, 4096, 4096), (16384, 4096, 4096), (27664, 4096, 4096), (27664, 22016, 4096),
          (27664, 4096, 11008)]
REPS = 3


def worker():
    import torch
    ops = importlib.import_module("rlaif-v_amd.ops")
    for M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        b = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(REPS):
            ops.gemm_nn(a, b, out=out)
        torch.cuda.synchronize()
        del a, b, out


def report(path):
    db = sqlite3.connect(path)
    cur = db.execute("select * from pmc_events limit 1")
    cols = [d[0] for d in cur.description]
    print("columns:", cols)
    tcol = "start" if "start" in cols else ("start_timestamp" if "start_timestamp" in cols else None)
    idcol = "dispatch_id" if "dispatch_id" in cols else None
    key = idcol or tcol
    rows = db.execute(f"select {key}, name, sum(counter_value), max(duration) from pmc_events where counter_name = 'FETCH_SIZE' "
                      f"and name like '%gemm_nn%' group by {key}, name order by {key}").fetchall()
    print(len(rows), "gemm_nn dispatches")
    i = 0
    for M, N, K in SHAPES:
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        ideal = tiles / 32 * 12 * 256 * K * 2
        alg = (M * K + K * N) * 2
        for _ in range(REPS):
            if i >= len(rows):
                return
            _, name, kb, dur = rows[i]
            i += 1
            b = 2.0 * kb * 1024.0          # x2: gfx950 FETCH_SIZE calibration for 16-B/lane reads
            print(f"M {M:6d} N {N:6d} K {K:6d}: {tiles:5d} tiles ({tiles / 256:5.1f} rounds)  fetch {b / 1e9:7.3f} GB  "
                  f"= {b / ideal:5.2f} x cluster-ideal, {b / alg:5.2f} x operand bytes;  {dur / 1e3:8.1f} us")


if __name__ == "__main__":
    worker() if sys.argv[1] == "worker" else report(sys.argv[2])
