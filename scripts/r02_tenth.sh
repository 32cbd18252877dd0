#!/bin/bash
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "f16x2 or split_bf16x3 or split_mirror" > gpurun_out/r02j_t_kernels.log 2>&1; tail -3 gpurun_out/r02j_t_kernels.log
timeout 900 python -m pytest tests/test_multiproc_gpu.py tests/test_parity_gpu.py -q -k "multiproc or ranks or bench or cold_diffusion or (short_trajectory and split3)" > gpurun_out/r02j_t_misc.log 2>&1; tail -5 gpurun_out/r02j_t_misc.log
timeout 600 python scripts/conv_sweep.py --precision 5 --reps 6 --batch 16 --vae-batch 16 --quick > gpurun_out/r02j_sweep_quick.txt 2>&1; tail -1 gpurun_out/r02j_sweep_quick.txt
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02j_bench.json 2> gpurun_out/r02j_bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02j_bench.json'))
print(d['value'], d['roofline']['families_ms'], [(a['conv_precision'], a['value']) for a in d['other_conv_arithmetic']])
PY
