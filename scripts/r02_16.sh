#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "fused_groupnorm or split_mirror or residual_bound_slots or test_conv_f16x2" > gpurun_out/r02p_t_kernels.log 2>&1; tail -3 gpurun_out/r02p_t_kernels.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path > gpurun_out/r02p_bench.json 2> gpurun_out/r02p_bench.err; python - <<PY
import json
d=json.load(open('gpurun_out/r02p_bench.json'))
print(d['value'], d['roofline']['families_ms'], d['roofline']['traffic'], d['roofline']['algorithmic_bytes_per_launch'])
PY
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "f16x2" > gpurun_out/r02p_t_parity.log 2>&1; tail -3 gpurun_out/r02p_t_parity.log
