#!/bin/bash
# round-6 session 3: the batch-general planner (fitted cost model + Winograd rule) on the GPU: (a) Winograd vs direct at B = 12 / 24 / 69 / 200 on the NEW planner,
# (b) the reference's bulk workload (chunks of 200, tail 69) again, (c) cfg2 must not move, (d) the planner / Winograd / parity tests that touch plans
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s3
mkdir -p $O
cd $R
timeout 900 python scripts/sample_dataset.py --synthetic --sample-batch 200 --n-samples 469 --steps-list 150 --labels No_Cardiomegaly:0 --no-files --compare-no-egress > $O/bulk200.txt 2>&1; tail -1 $O/bulk200.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path --no-other-workloads > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
for b in 200 69 24 12; do
  timeout 1200 python scripts/wino_sweep.py --batch $b --reps 4 --cold-mb 600 > $O/wino_sweep_b$b.txt 2>&1; tail -2 $O/wino_sweep_b$b.txt
done
timeout 1500 python -m pytest tests/test_winograd_gpu.py tests/test_kernels_gpu.py -x -q > $O/tests_kernels.txt 2>&1; tail -3 $O/tests_kernels.txt
