#!/bin/bash
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "test_conv_f16x2 and (61 or 63)" > gpurun_out/r02n_t_kernels.log 2>&1; tail -5 gpurun_out/r02n_t_kernels.log
timeout 1200 python scripts/conv_sweep.py --precision 5 --reps 6 --batch 16 --vae-batch 16 --tiles 31,33,52,53,61,63 > gpurun_out/r02n_sweep.txt 2>&1; tail -45 gpurun_out/r02n_sweep.txt
