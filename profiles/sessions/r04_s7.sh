#!/bin/bash
# (needs the library of commit a578755: MF_CONV_TREE=3 was reverted after this session -- profiles/r04_tree_first_arriver_ab.txt)
# round-4 GPU session 7: split-K hand-off with the pair's counter FIRST (MF_CONV_TREE=3: only the first arriver stores): correctness, stress, A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s7
mkdir -p $O
cd $R
MF_CONV_TREE=3 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "test_conv_f16x2 and not stress and not cancelling" > $O/tests_tree3.txt 2>&1; tail -3 $O/tests_tree3.txt
MF_CONV_TREE=3 timeout 900 python scripts/conv_stress.py --reps 2000 > $O/stress_tree3.txt 2>&1; tail -6 $O/stress_tree3.txt
for round in 1 2 3; do
  for t in 1 3; do
    MF_CONV_TREE=$t timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt-path --no-other-workloads 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MF_CONV_TREE=$t round $round:', d['value'], d['ms_per_step'], d['roofline']['families_ms']['conv_igemm'], d['roofline']['frac'])"
  done
done
