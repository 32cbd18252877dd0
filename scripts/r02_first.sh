#!/bin/bash
# first GPU session of round 2: the fp16-pair LDS-DMA conv kernel -- parity tests, per-shape sweep, same-session comparison, end-to-end
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "f16x2 or split_mirror" > gpurun_out/r02_t_f16x2.log 2>&1; tail -5 gpurun_out/r02_t_f16x2.log
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 6 > gpurun_out/r02_sweep_p5.txt 2>&1; tail -3 gpurun_out/r02_sweep_p5.txt
timeout 300 python scripts/conv_sweep.py --precision 3 --reps 6 --quick > gpurun_out/r02_sweep_p3_quick.txt 2>&1; tail -2 gpurun_out/r02_sweep_p3_quick.txt
MEDFUSION_CONV_PRECISION=5 timeout 900 python -m pytest tests/test_parity_gpu.py -x -q > gpurun_out/r02_t_parity_p5.log 2>&1; tail -5 gpurun_out/r02_t_parity_p5.log
timeout 600 python bench.py --steps 2 --warmup 1 --conv-precision 5 --no-cpu-baseline > gpurun_out/r02_bench_p5.json 2> gpurun_out/r02_bench_p5.err; cat gpurun_out/r02_bench_p5.json; tail -3 gpurun_out/r02_bench_p5.err
