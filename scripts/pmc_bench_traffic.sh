#!/bin/bash
# HBM traffic of the bench's kernels: rocprofv3 PMC passes over ONE step of the bench command (FETCH_SIZE and WRITE_SIZE in
# separate passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes), summarised per kernel and per launch.
# usage (GPU box): scripts/pmc_bench_traffic.sh gpurun_out/pmc_traffic   ->  <out>/traffic.json  (copy to profiles/pmc_bench_traffic.json)
OUT=$1; R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 1200 rocprofv3 --pmc $c --kernel-trace -d $R/$OUT -o $c --output-format csv -- \
    python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alt-path --no-roofline --no-other-workloads > $R/$OUT/run_$c.log 2>&1
done
python - "$R/$OUT" "$R" <<'PY'
import collections, csv, json, sys
d = sys.argv[1]
sys.path.insert(0, sys.argv[2])
from medfusion_amd.build import conv_source_stamp
tab = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{d}/{c}_counter_collection.csv")):
        if r["Counter_Name"] == c:
            tab[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
out = []
for k, v in tab.items():
    if not v["FETCH_SIZE"]:
        continue
    f = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
    w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
    # units: KiB.  gfx950: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes -> x2 (guide, HBM section)
    out.append({"kernel": k[:160], "launches": len(v["FETCH_SIZE"]), "fetch_KiB_avg_raw": round(f, 1), "write_KiB_avg_raw": round(w, 1),
                "hbm_bytes_per_launch": int((2 * f + w) * 1024), "total_fetch_KiB_raw": round(sum(v["FETCH_SIZE"]), 1)})
out.sort(key=lambda e: -e["total_fetch_KiB_raw"])
# per-tile breakdown of the conv kernel's fetches against what ONE pass over its operands would move: the bench's conv
# launches by tile instantiation, fetched bytes / launch next to written bytes / launch
tiles = [{"tile": e["kernel"][e["kernel"].index("<") + 1:e["kernel"].rindex(">")], "launches": e["launches"], "fetch_MB_per_launch": round(2 * e["fetch_KiB_avg_raw"] * 1024 / 1e6, 2),
          "write_MB_per_launch": round(e["write_KiB_avg_raw"] * 1024 / 1e6, 2)} for e in out if "conv_f16x2_kernel<" in e["kernel"] or "conv_halo_kernel<" in e["kernel"] or "conv_group_kernel<" in e["kernel"]]
json.dump({"command": "bench.py --steps 1 --warmup 0 (cfg2: B=16, 150 iterations + decode, default conv arithmetic)",
           "conv_source_stamp": conv_source_stamp(), "conv_tiles": tiles,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction)",
           "kernels": out[:24]}, open(f"{d}/traffic.json", "w"), indent=1)
for e in out[:8]:
    print(e["launches"], e["hbm_bytes_per_launch"] / 1e6, "MB/launch", e["kernel"][:90])
PY
