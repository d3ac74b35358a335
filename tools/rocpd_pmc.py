"""Per-kernel average of a rocprofv3 --pmc counter from the rocpd SQLite output.
Usage: python tools/rocpd_pmc.py results.db [kernel-substring]"""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(anonymous namespace\)::|void\s+", "", name)[:90]


def summarize(path, substr=None):
    db = sqlite3.connect(path)
    rows = db.execute("select name, counter_name, count(*), sum(counter_value), avg(counter_value), sum(duration) "
                      "from pmc_events group by name, counter_name order by sum(counter_value) desc").fetchall()
    out = []
    for n, c, cnt, tot, avg, dur in rows:
        if substr and substr not in n:
            continue
        out.append(dict(kernel=short(n), counter=c, calls=cnt, total=tot, avg=avg, total_duration_ns=dur))
    return out


GROUPS = ["gemm_nn_a64_kernel<EpiStore", "gemm_nn_256_kernel<EpiStore", "gemm_nt_256_kernel<EpiStore", "gemm_tn_256_kernel<EpiStore",
          "gemm_nn_a64_kernel<EpiSwiGLU,", "gemm_nn_a64_kernel<EpiSwiGLUBwd", "gemm_nn_a64_kernel<EpiStoreRope", "gemm_nt_256_kernel<EpiLogpFwd", "gemm_nt_256_kernel<EpiLogpBwd",
          "attn_fwd2_kernel<128", "attn_fwd3_kernel", "attn_bwd_dq2", "attn_bwd_dkv5", "adamw_kernel", "swiglu_fwd", "rmsnorm_fwd"]


def traffic_json(fetch_db, write_db):
    """Per kernel group: average FETCH_SIZE / WRITE_SIZE (KB per launch) and the corrected HBM bytes per launch
    (2 x FETCH + WRITE on gfx950, MI355X_MICROARCH.md HBM section)."""
    import json
    f = summarize(fetch_db)
    w = summarize(write_db)
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 1 "
                      "--no-cpu-baseline --no-gemm-timer (two separate passes; tools/collect_pmc_traffic.sh)",
           "units": "KB per launch (counter definition); FETCH_SIZE doubled on gfx950 for 16-B/lane coalesced reads "
                    "(MI355X_MICROARCH.md HBM section; sanity: adamw_kernel reads 12 B + writes 14 B per parameter)",
           "kernels": {}}
    tot_b, tot_c = 0.0, 0
    for g in GROUPS:
        fr = [r for r in f if g in r["kernel"] and r["counter"] == "FETCH_SIZE"]
        wr = [r for r in w if g in r["kernel"] and r["counter"] == "WRITE_SIZE"]
        if not fr or not wr:
            continue
        calls = sum(r["calls"] for r in fr)
        fk = sum(r["total"] for r in fr) / calls
        wk = sum(r["total"] for r in wr) / max(sum(r["calls"] for r in wr), 1)
        b = (2.0 * fk + wk) * 1024.0
        out["kernels"][g] = {"calls": calls, "fetch_kb_avg": fk, "write_kb_avg": wk, "hbm_bytes_per_launch_corrected": b}
        if g.startswith("gemm_"):
            tot_b += b * calls
            tot_c += calls
    out["gemm_all_launches_hbm_bytes_per_launch"] = tot_b / max(tot_c, 1)
    return json.dumps(out, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "--json":
        print(traffic_json(sys.argv[2], sys.argv[3]))
        sys.exit(0)
    for r in summarize(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)[:80]:
        print(f"{r['kernel']:92s} {r['counter']:10s} calls {r['calls']:5d}  avg {r['avg']:14.1f}  total {r['total']:16.1f}")
