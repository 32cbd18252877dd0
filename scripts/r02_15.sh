#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
timeout 600 python scripts/enqueue_time.py 2>&1 | tail -8 | tee gpurun_out/r02o_enqueue.txt
