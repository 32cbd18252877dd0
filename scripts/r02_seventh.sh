#!/bin/bash
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 600 python bench.py --steps 2 --warmup 1 --conv-precision 5 --no-cpu-baseline --alt-precision 1 > gpurun_out/r02g_bench_p5.json 2> gpurun_out/r02g_bench_p5.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02g_bench_p5.json'))
print(d['value'], d['roofline']['families_ms'], d['other_conv_arithmetic'])
PY
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r02g_t_all.log 2>&1; tail -15 gpurun_out/r02g_t_all.log
