#!/bin/bash
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "fused_groupnorm or split_mirror" > gpurun_out/r02m_t_kernels.log 2>&1; tail -3 gpurun_out/r02m_t_kernels.log
for fin in 1 0 1 0; do
MEDFUSION_FINALIZE_IN_APPLY=$fin timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path > gpurun_out/r02m_bench_fin$fin.json 2> gpurun_out/r02m_bench_fin$fin.err; python - <<PY
import json
d=json.load(open('gpurun_out/r02m_bench_fin$fin.json'))
print($fin, d['value'], d['roofline']['families_ms'])
PY
done
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "f16x2" > gpurun_out/r02m_t_parity.log 2>&1; tail -3 gpurun_out/r02m_t_parity.log
