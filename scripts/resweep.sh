#!/bin/bash
# regenerate the planner table of the fp16-pair convolution: sweeps at the batch sizes / latents of the BASELINE configs, every shape where the
# sweep beats the planner's own choice by more than the noise goes to gpurun_out/plan_table_*.inc (merge into csrc/conv_plan_table.inc)
cd /root/repo; mkdir -p gpurun_out; rm -f gpurun_out/plan_table_*.inc
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 6 --batch 16 --vae-batch 16 --emit-table gpurun_out/plan_table_b16.inc > gpurun_out/r02_resweep_b16.txt 2>&1; tail -1 gpurun_out/r02_resweep_b16.txt
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 6 --batch 32 --vae-batch 16 --emit-table gpurun_out/plan_table_b32.inc > gpurun_out/r02_resweep_b32.txt 2>&1; tail -1 gpurun_out/r02_resweep_b32.txt
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 6 --batch 8 --vae-batch 8 --emit-table gpurun_out/plan_table_b8.inc > gpurun_out/r02_resweep_b8.txt 2>&1; tail -1 gpurun_out/r02_resweep_b8.txt
timeout 900 python scripts/conv_sweep.py --precision 5 --reps 6 --batch 8 --vae-batch 8 --latent 64 --emit-table gpurun_out/plan_table_l64.inc > gpurun_out/r02_resweep_l64.txt 2>&1; tail -1 gpurun_out/r02_resweep_l64.txt
