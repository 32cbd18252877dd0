#!/usr/bin/env python3
"""Average the per-dispatch counter values of rocprofv3 --pmc CSV outputs for the conv_igemm kernel."""
import collections, csv, glob, sys
d = sys.argv[1]
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "conv_igemm"
out = []
for f in sorted(glob.glob(f"{d}/pmc*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = v[5:] if len(v) > 6 else v
        out.append(f"{k},{sum(v) / len(v):.1f}")
for f in sorted(glob.glob(f"{d}/pmc1_kernel_trace.csv")):
    t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"]]
    t = t[5:] if len(t) > 6 else t
    out.append(f"kernel_us_under_pmc,{sum(t) / len(t):.2f}")
print("\n".join(out))
open(f"{d}/summary.csv", "w").write("\n".join(out) + "\n")
