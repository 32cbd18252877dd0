#!/bin/bash
# round-4 GPU session 1: the new parity tests with their measured values, the baseline bench line (all BASELINE configs), the attention
# numbers, the VAE shapes at the batch sample() decodes.  Everything lands in gpurun_out/r04s1/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s1
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -s -x -k "cancelling or learned or rccl_world1 or test_conv_direct or unit_range or full_length or benchmarked_batch" > $O/tests_new.txt 2>&1; tail -5 $O/tests_new.txt
grep -h "measured\|rel-err\|RCCL1_OK\|unit-range\|cfg2 batch rows" $O/tests_new.txt > $O/tests_new_measured.txt
timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; tail -3 $O/bench.err
timeout 300 python scripts/attention_bench.py > $O/attention.txt 2>&1; cat $O/attention.txt
timeout 600 python scripts/conv_sweep.py --precision 5 --reps 6 --batch 16 --vae-batch 16 --only vae --emit-table $O/vae_b16_table.inc > $O/sweep_vae_b16.txt 2>&1; tail -12 $O/sweep_vae_b16.txt
timeout 600 python scripts/conv_sweep.py --precision 5 --reps 6 --batch 8 --vae-batch 8 --latent 64 --only vae --emit-table $O/vae_l64_table.inc > $O/sweep_vae_l64.txt 2>&1; tail -12 $O/sweep_vae_l64.txt
