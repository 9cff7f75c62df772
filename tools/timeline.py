#!/usr/bin/env python
"""Timeline of the last N kernel dispatches from a rocprofv3 rocpd database: start offset (us),
duration (us), gap to the previous kernel's end, queue, short name.
usage: tools/timeline.py results.db [N] [out.txt]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    n = n.replace("at::native::", "")
    return n[:70]


def main():
    db = sys.argv[1]
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    qcol = next((x for x in ("queue_id", "stream_id", "queue", "stream") if x in cols), None)
    sel = f"select start, end, {name_col}" + (f", {qcol}" if qcol else ", 0") + \
          f" from kernels order by start desc limit {N}"
    rows = list(reversed(c.execute(sel).fetchall()))
    t0 = rows[0][0]
    out = [f"# columns in kernels view: {cols}"]
    last_end = t0
    for s, e, n, q in rows:
        out.append(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - last_end) / 1e3:7.1f}"
                   f"  q={q}  {short(n)}")
        last_end = max(last_end, e)
    text = "\n".join(out)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
