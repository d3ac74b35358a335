"""A/B of GEMM schedule variants (tools/build_gemm_variants.py) and a power / clock probe.

    python tools/exp_gemm_variants.py ab        # every variant library, two alternating passes, NN + TN step shapes
    python tools/exp_gemm_variants.py power     # socket power + shader clock while our gate_up GEMM / hipBLASLt loop runs

Each variant runs in its own process (RV_HIP_LIB selects the library); quote deltas only from the same call."""
import glob
import json
import os
import subprocess
import sys
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def worker():
    import torch
    from rlaif_v_amd import ops
    BF = torch.bfloat16
    R = 27664

    def timeit(fn, iters=8, warmup=2):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters

    res = {}
    for name, N, K in [("nn_gate_up", 22016, 4096), ("nn_down", 4096, 11008), ("nn_qkv", 12288, 4096)]:
        x = torch.randn(R, K, device="cuda").to(BF)
        wT = (torch.randn(K, N, device="cuda") * 0.02).to(BF)
        out = torch.empty(R, N, device="cuda", dtype=BF)
        ms = timeit(lambda: ops.gemm_nn(x, wT, out=out))
        res[name] = 2.0 * R * N * K / ms / 1e9
        del x, wT, out
    for name, I, J in [("tn_wgu", 22016, 4096), ("tn_wqkv", 12288, 4096)]:
        p = torch.randn(R, I, device="cuda").to(BF)
        q = torch.randn(R, J, device="cuda").to(BF)
        out = torch.empty(I, J, device="cuda", dtype=BF)
        ms = timeit(lambda: ops.gemm_tn(p, q, out=out))
        res[name] = 2.0 * R * I * J / ms / 1e9
        del p, q, out
    print("RESULT " + json.dumps(res), flush=True)


def ab():
    libs = {"default": os.path.join(REPO, "rlaif-v_amd", "librlaifv_hip.so")}
    for f in sorted(glob.glob(os.path.join(REPO, "rlaif-v_amd", "librlaifv_hip_*.so"))):
        libs[os.path.basename(f)[len("librlaifv_hip_"):-3]] = f
    acc = {k: [] for k in libs}
    for rnd in range(2):
        order = list(libs) if rnd == 0 else list(reversed(list(libs)))
        for k in order:
            env = dict(os.environ, RV_HIP_LIB=libs[k])
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(k, "FAILED", out.stderr[-500:])
                continue
            acc[k].append(json.loads(line[0][7:]))
    names = list(acc["default"][0]) if acc["default"] else []
    print(f"{'variant':10s} " + " ".join(f"{n:>12s}" for n in names) + "   (TF/s, mean of passes; relative to default)")
    base = {n: sum(r[n] for r in acc["default"]) / len(acc["default"]) for n in names}
    for k, rs in acc.items():
        if not rs:
            continue
        m = {n: sum(r[n] for r in rs) / len(rs) for n in names}
        print(f"{k:10s} " + " ".join(f"{m[n]:7.0f}{100 * (m[n] / base[n] - 1):+5.1f}%" for n in names))


def sample_smi(stop, rows):
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out)
            c = d.get("card0", {})
            rows.append({k: v for k, v in c.items() if "ower" in k or "sclk" in k or "mclk" in k or "fclk" in k})
        except Exception as e:       # noqa: BLE001
            rows.append({"error": repr(e)[:100]})
        time.sleep(0.05)


def power():
    import torch
    from rlaif_v_amd import ops
    BF = torch.bfloat16
    R, N, K = 27664, 22016, 4096
    x = torch.randn(R, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    wT = w.t().contiguous()
    out = torch.empty(R, N, device="cuda", dtype=BF)
    p = torch.randn(R, N, device="cuda").to(BF)
    gw = torch.empty(N, K, device="cuda", dtype=BF)
    cases = [("ours nn gate_up", lambda: ops.gemm_nn(x, wT, out=out)), ("hipBLASLt x @ w.T", lambda: torch.matmul(x, w.t(), out=out)),
             ("ours tn wgu", lambda: ops.gemm_tn(p, x, out=gw)), ("idle", None)]
    for name, fn in cases:
        rows, stop = [], threading.Event()
        th = threading.Thread(target=sample_smi, args=(stop, rows))
        t0 = time.time()
        n = 0
        th.start()
        if fn is None:
            time.sleep(3)
        else:
            while time.time() - t0 < 4.0:
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                n += 20
        dt = time.time() - t0
        stop.set()
        th.join()
        tf = 2.0 * R * N * K * n / dt / 1e12 if fn else 0.0
        print(f"== {name}: {tf:.0f} TF/s sustained over {dt:.1f} s; samples:")
        for r in rows[1:6]:
            print("   ", json.dumps(r))


if __name__ == "__main__":
    {"worker": worker, "ab": ab, "power": power}[sys.argv[1]]()
