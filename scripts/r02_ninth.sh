#!/bin/bash
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r02i_t_all.log 2>&1; tail -8 gpurun_out/r02i_t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r02i_bench.json 2> gpurun_out/r02i_bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02i_bench.json'))
print(d['value'], d['roofline']['families_ms'], d['other_conv_arithmetic'], d['cpu_baseline'])
PY
