#!/bin/bash
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 300 python scripts/split_accuracy.py > gpurun_out/r02h_split_accuracy.txt 2>&1; cat gpurun_out/r02h_split_accuracy.txt
: > gpurun_out/r02h_table.inc
timeout 600 python scripts/conv_sweep.py --precision 5 --reps 6 --batch 16 --vae-batch 16 --emit-table gpurun_out/r02h_table.inc > gpurun_out/r02h_sweep_b16.txt 2>&1; tail -1 gpurun_out/r02h_sweep_b16.txt
timeout 600 python scripts/conv_sweep.py --precision 5 --reps 6 --batch 32 --vae-batch 8 --emit-table gpurun_out/r02h_table.inc > gpurun_out/r02h_sweep_b32.txt 2>&1; tail -1 gpurun_out/r02h_sweep_b32.txt
timeout 600 python scripts/conv_sweep.py --precision 5 --reps 6 --batch 8 --vae-batch 4 --emit-table gpurun_out/r02h_table.inc > gpurun_out/r02h_sweep_b8.txt 2>&1; tail -1 gpurun_out/r02h_sweep_b8.txt
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 4 --batch 8 --vae-batch 8 --latent 64 --emit-table gpurun_out/r02h_table.inc > gpurun_out/r02h_sweep_l64.txt 2>&1; tail -1 gpurun_out/r02h_sweep_l64.txt
wc -l gpurun_out/r02h_table.inc
timeout 600 python bench.py --steps 2 --warmup 1 --conv-precision 5 --no-cpu-baseline --alt-precision 1 > gpurun_out/r02h_bench_p5.json 2> gpurun_out/r02h_bench_p5.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02h_bench_p5.json'))
print(d['value'], d['roofline']['families_ms'], d['other_conv_arithmetic'])
PY
