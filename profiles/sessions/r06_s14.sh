#!/bin/bash
# round-6 session 14: the Winograd form on the exact bf16-triplet arithmetic through the whole pipeline: parity tests on that arithmetic (direct and Winograd
# everywhere), then the cfg2 step: direct (what other_conv_arithmetic reported until round 5) / the pair arithmetic's shape rule / wherever the library can
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s14
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_parity_gpu.py -q -x -s -k "split3" > $O/tests_split3.txt 2>&1; tail -5 $O/tests_split3.txt
timeout 1800 python scripts/env_ab.py --rounds 2 "MEDFUSION_CONV_PRECISION=1 MEDFUSION_WINOGRAD_F32=0" "MEDFUSION_CONV_PRECISION=1 MEDFUSION_WINOGRAD_F32=1" "MEDFUSION_CONV_PRECISION=1 MEDFUSION_WINOGRAD_F32=2" > $O/ab_f32.txt 2>&1; cat $O/ab_f32.txt
