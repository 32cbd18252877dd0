#!/bin/bash
# GPU session 3 of round 2: per-sample operand scaling + deeper LDS pipeline of the fp16-pair conv
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "f16x2 or split_mirror" > gpurun_out/r02c_t_f16x2.log 2>&1; tail -8 gpurun_out/r02c_t_f16x2.log
timeout 1500 python -m pytest tests/test_parity_gpu.py -q -s -k "f16x2" > gpurun_out/r02c_t_parity_p5.log 2>&1; grep -E "passed|failed|conv precision|FAILED" gpurun_out/r02c_t_parity_p5.log | tail -12
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 6 --tiles 31,32,33,34,35,36,37,43,46 > gpurun_out/r02c_sweep_p5.txt 2>&1; tail -3 gpurun_out/r02c_sweep_p5.txt
timeout 600 python bench.py --steps 2 --warmup 1 --conv-precision 5 --no-cpu-baseline --alt-precision 1 > gpurun_out/r02c_bench_p5.json 2> gpurun_out/r02c_bench_p5.err; cat gpurun_out/r02c_bench_p5.json; tail -3 gpurun_out/r02c_bench_p5.err
