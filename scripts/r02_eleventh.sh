#!/bin/bash
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "f16x2 or split_bf16x3 or split_mirror or fused_groupnorm" > gpurun_out/r02k_t_kernels.log 2>&1; tail -3 gpurun_out/r02k_t_kernels.log
timeout 900 python -m pytest tests/test_multiproc_gpu.py tests/test_parity_gpu.py -q -k "multiproc or ranks or bench or cold_diffusion or short_trajectory" > gpurun_out/r02k_t_misc.log 2>&1; tail -5 gpurun_out/r02k_t_misc.log
timeout 600 python scripts/conv_sweep.py --precision 5 --reps 6 --batch 16 --vae-batch 16 --quick > gpurun_out/r02k_sweep_quick.txt 2>&1; tail -1 gpurun_out/r02k_sweep_quick.txt
for fin in 1 0; do
MEDFUSION_FINALIZE_IN_APPLY=$fin timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-precision 1 > gpurun_out/r02k_bench_fin$fin.json 2> gpurun_out/r02k_bench_fin$fin.err; python - <<PY
import json
d=json.load(open('gpurun_out/r02k_bench_fin$fin.json'))
print($fin, d['value'], d['roofline']['families_ms'], [(a['conv_precision'], a['value']) for a in d['other_conv_arithmetic']])
PY
done
