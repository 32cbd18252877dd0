#!/usr/bin/env python3
"""Where the wall time of the denoise loop goes that is NOT inside a kernel: reads a rocprofv3 --kernel-trace CSV
(Start_Timestamp / End_Timestamp per dispatch), sorts the dispatches of the device by start time and reports the sum of
durations, the span, the idle time between consecutive dispatches (overall and by the kernel that FOLLOWS the gap), and
per-kernel totals.  usage: trace_gaps.py <..._kernel_trace.csv> [out.txt]"""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n).split("(")[0]
    return n if len(n) < 90 else n[:87] + "..."


def main(path, out=None):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # the steady part: drop everything before the first long idle stretch after start-up (> 50 ms: weight packing, warm-up sync)
    lines = [f"# {path}: {len(rows)} dispatches"]
    segs, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if b[0] - a[1] > 20_000_000:
            segs.append(cur)
            cur = []
        cur.append(b)
    segs.append(cur)
    seg = max(segs, key=len)
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [(b[0] - a[1], b[2]) for a, b in zip(seg, seg[1:])]
    idle = sum(max(0, g) for g, _ in gaps)
    lines.append(f"largest contiguous segment: {len(seg)} dispatches, span {span / 1e6:.2f} ms, in kernels {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), "
                 f"idle between dispatches {idle / 1e6:.2f} ms ({100 * idle / span:.1f} %), mean gap {idle / max(1, len(gaps)) / 1e3:.2f} us")
    hist = collections.Counter()
    for g, _ in gaps:
        hist["<1us" if g < 1000 else "1-2us" if g < 2000 else "2-4us" if g < 4000 else "4-8us" if g < 8000 else "8-20us" if g < 20000 else ">20us"] += 1
    lines.append("gap histogram: " + ", ".join(f"{k}: {hist[k]}" for k in ("<1us", "1-2us", "2-4us", "4-8us", "8-20us", ">20us")))
    by = collections.defaultdict(lambda: [0, 0, 0])
    for (g, nm), (s, e, _) in zip(gaps, seg[1:]):
        by[nm][0] += 1
        by[nm][1] += max(0, g)
        by[nm][2] += e - s
    lines.append("kernel,calls,total_ms,avg_us,idle_before_total_ms,idle_before_avg_us")
    for nm, (c, g, d) in sorted(by.items(), key=lambda kv: -kv[1][2]):
        lines.append(f"\"{nm}\",{c},{d / 1e6:.3f},{d / c / 1e3:.2f},{g / 1e6:.3f},{g / c / 1e3:.2f}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
