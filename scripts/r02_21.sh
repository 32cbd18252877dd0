#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path > gpurun_out/r02u_bench.json 2> gpurun_out/r02u_bench.err; python - <<PY
import json
d=json.load(open('gpurun_out/r02u_bench.json'))
r=d['roofline']
print(d['value'], r['achieved'], r['frac'], r['mfma_sustained'], r['frac_of_sustained_random_operands'])
PY
tail -3 gpurun_out/r02u_bench.err
