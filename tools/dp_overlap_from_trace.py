"""Does the data-parallel stand-in kernel really run BESIDE the backward GEMMs?  Reads a rocprofv3 --kernel-trace rocpd
database of `bench.py --dp-probe-only` and reports, for every reduce_copy_persistent_kernel launch, how much of its interval
is covered by other kernels (and by which).  Usage: python tools/dp_overlap_from_trace.py results.db [out.json]"""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    s_col = "start" if "start" in cols else "start_timestamp"
    e_col = "end" if "end" in cols else "end_timestamp"
    rows = db.execute(f"select name, {s_col}, {e_col} from kernels order by {s_col}").fetchall()
    comm = [(s, e) for n, s, e in rows if "reduce_copy_persistent" in n]
    other = [(s, e, re.sub(r"<.*", "", re.sub(r"void\s+|\(anonymous namespace\)::", "", n))[:40]) for n, s, e in rows
             if "reduce_copy_persistent" not in n]
    out = dict(comm_launches=len(comm), comm_total_ms=sum(e - s for s, e in comm) / 1e6)
    covered, by = 0, defaultdict(int)
    j0 = 0
    for s, e in comm:
        while j0 < len(other) and other[j0][1] <= s:
            j0 += 1
        j, cur = j0, s
        while j < len(other) and other[j][0] < e:
            os_, oe, on = other[j]
            a, b = max(os_, cur), min(oe, e)
            if b > a:                      # other kernels of one stream do not overlap each other: simple sweep
                covered += b - a
                by[on] += b - a
                cur = b
            j += 1
    out["comm_ms_covered_by_other_kernels"] = covered / 1e6
    out["covered_frac"] = covered / max(sum(e - s for s, e in comm), 1)
    out["covered_by_ms"] = {k: round(v / 1e6, 2) for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]}
    out["comm_avg_GBps_local_traffic_note"] = "3 x bucket bytes (two loads + one store per element) / kernel duration"
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
