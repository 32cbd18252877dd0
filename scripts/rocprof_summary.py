#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results .db (rocpd sqlite) into a small per-kernel table
(name, calls, total/avg/min/max duration, share) -- the form committed under profiles/."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"# rocprofv3 --kernel-trace summary of {path}", f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches",
             "name,calls,total_ms,avg_us,min_us,max_us,percent"]
    for n, c, s, a, mn, mx in rows:
        short = n if len(n) < 140 else n[:137] + "..."
        lines.append(f"\"{short}\",{c},{s / 1e6:.3f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * s / tot:.2f}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
