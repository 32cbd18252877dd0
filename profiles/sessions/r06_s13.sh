#!/bin/bash
# round-6 session 13: the Winograd form on the exact arithmetics (ABI 250): kernel-level tests, then the arithmetic-1 step with and without it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s13
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_winograd_gpu.py -q -x -s -k "exact_arithmetics or f32_refuses" > $O/tests_f32.txt 2>&1; tail -12 $O/tests_f32.txt
