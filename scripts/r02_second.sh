#!/bin/bash
# GPU session 2 of round 2: kernel-level parity of the fp16-pair conv, the whole parity suite on it, accuracy table, PMC of the kernel
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "f16x2 or split_mirror" > gpurun_out/r02_t_f16x2.log 2>&1; tail -8 gpurun_out/r02_t_f16x2.log
timeout 300 python scripts/split_accuracy.py > gpurun_out/r02_split_accuracy.txt 2>&1; cat gpurun_out/r02_split_accuracy.txt
timeout 1500 python -m pytest tests/test_parity_gpu.py -q -s -k "f16x2" > gpurun_out/r02_t_parity_p5.log 2>&1; grep -E "passed|failed|conv precision" gpurun_out/r02_t_parity_p5.log | tail -8
bash scripts/pmc_conv.sh gpurun_out/r02_pmc_t33 33 1 16,32,32,256,0,256,3,1,0 all 5 | tail -60
bash scripts/pmc_conv.sh gpurun_out/r02_pmc_t31 31 8 16,8,8,1024,1024,1024,3,1,0 all 5 | tail -60
