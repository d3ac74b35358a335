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


if __name__ == "__main__":
    for r in summarize(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)[:25]:
        print(f"{r['kernel']:92s} {r['counter']:10s} calls {r['calls']:5d}  avg {r['avg']:14.1f}  total {r['total']:16.1f}")
