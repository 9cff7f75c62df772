#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max ns, % of GPU kernel time) from a rocprofv3
rocpd sqlite database -- the same table `rocprofv3 --stats` prints, written as CSV/markdown so it
can be committed under profiles/.   usage: tools/rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), "
        f"max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["name,calls,total_ns,avg_ns,min_ns,max_ns,pct"]
    for n, k, t, a, lo, hi in rows:
        lines.append(f"\"{n}\",{k},{t},{a:.0f},{lo},{hi},{100.0 * t / tot:.2f}")
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
