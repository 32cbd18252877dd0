#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_harness_gpu.py -q -x -k "graph" > gpurun_out/r02r_t_graph.log 2>&1; tail -3 gpurun_out/r02r_t_graph.log
timeout 600 python scripts/enqueue_time.py 2>&1 | tail -10 | tee gpurun_out/r02r_enqueue.txt
