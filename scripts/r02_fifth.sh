#!/bin/bash
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "f16x2 or split_mirror" > gpurun_out/r02e_t_f16x2.log 2>&1; tail -4 gpurun_out/r02e_t_f16x2.log
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 6 > gpurun_out/r02e_sweep_p5.txt 2>&1; tail -3 gpurun_out/r02e_sweep_p5.txt
timeout 600 python bench.py --steps 2 --warmup 1 --conv-precision 5 --no-cpu-baseline --alt-precision 1 > gpurun_out/r02e_bench_p5.json 2> gpurun_out/r02e_bench_p5.err; cat gpurun_out/r02e_bench_p5.json; tail -3 gpurun_out/r02e_bench_p5.err
timeout 1200 python -m pytest tests/test_harness_gpu.py -q -x > gpurun_out/r02e_t_harness.log 2>&1; tail -15 gpurun_out/r02e_t_harness.log
