"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace: per-kernel calls / total / average
duration, the same table `--stats` prints as CSV.  Usage: python tools/rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void\s+", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for n, c, t, a, mn, mx in rows:
        lines.append(f"\"{short(n)}\",{c},{t},{a:.0f},{mn},{mx},{100.0 * t / total:.2f}")
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
