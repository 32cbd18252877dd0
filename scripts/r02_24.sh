#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "f16x2" > gpurun_out/r02x_t_parity.log 2>&1; tail -3 gpurun_out/r02x_t_parity.log
timeout 600 python scripts/enqueue_time.py 2>&1 | tail -10 | tee gpurun_out/r02x_enqueue.txt
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path --no-roofline | head -c 300
