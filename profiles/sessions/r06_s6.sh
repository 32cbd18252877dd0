#!/bin/bash
# round-6 session 6: the full GPU suite on the batch-general planner (direct model, GEMM model, Winograd rule), the bulk workload on it, cfg2,
# and cfg4's batch with the block epilogue inside the convolution's launch (MEDFUSION_FUSED_APPLY=1: measured neutral at B = 16 in round 4; B = 8?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s6
mkdir -p $O
cd $R
timeout 900 python scripts/sample_dataset.py --synthetic --sample-batch 200 --n-samples 869 --steps-list 150 --labels No_Cardiomegaly:0 --no-files --compare-no-egress > $O/bulk200.txt 2>&1; tail -1 $O/bulk200.txt
timeout 900 python scripts/env_ab.py --rounds 2 --batch 8 "" "MEDFUSION_FUSED_APPLY=1" > $O/ab_b8_fused.txt 2>&1; cat $O/ab_b8_fused.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests_all.txt 2>&1; tail -5 $O/tests_all.txt
