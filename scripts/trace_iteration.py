#!/usr/bin/env python3
"""The launches of ONE denoise iteration in issue order with their durations, from a rocprofv3 --kernel-trace CSV: the dispatches between the
last two sched_step_philox kernels.  usage: trace_iteration.py <..._kernel_trace.csv> [out.txt]"""
import csv, re, sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("mfc2::", "").replace("mfw::", "")
    n = re.sub(r"^void ", "", n).split("(")[0]
    return n if len(n) < 100 else n[:97] + "..."


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    g = r.get("Grid_Size") or r.get("Grid_Size_X") or ""
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), g, r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or ""))
rows.sort()
idx = [i for i, r in enumerate(rows) if "sched_step_philox" in r[2]]
a, b = idx[-2], idx[-1]
out = [f"# one iteration: dispatches {a + 1} .. {b} of {sys.argv[1]}", "start_us  dur_us  gap_us  grid  kernel"]
t0 = rows[a][1]
prev = rows[a][1]
tot = 0
for s, e, n, g, w in rows[a + 1:b + 1]:
    out.append(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {(s - prev) / 1e3:6.1f}  {g:>7s} {n}")
    prev = e
    tot += e - s
out.append(f"# span {(rows[b][1] - t0) / 1e3:.1f} us, in kernels {tot / 1e3:.1f} us")
txt = "\n".join(out) + "\n"
open(sys.argv[2], "w").write(txt) if len(sys.argv) > 2 else print(txt)
