#!/bin/bash
# round-6 session 7: the bulk workload WITH PNG files (2 encoder threads) -> profiles/r06_bulk_generation.json; the tests added since session 6
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s7
mkdir -p $O
cd $R
timeout 900 python scripts/sample_dataset.py --synthetic --sample-batch 200 --n-samples 1069 --steps-list 150 --labels No_Cardiomegaly:0 --compare-no-egress --out /tmp/r06_generated > $O/bulk200_png.txt 2>&1; tail -1 $O/bulk200_png.txt; ls /tmp/r06_generated_150/No_Cardiomegaly | wc -l
timeout 900 python scripts/sample_dataset.py --synthetic --sample-batch 200 --n-samples 469 --steps-list 150 --labels No_Cardiomegaly:0 --compare-no-egress --writer-threads 1 --out /tmp/r06_generated1 > $O/bulk200_png_1thread.txt 2>&1; tail -1 $O/bulk200_png_1thread.txt
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_winograd_gpu.py tests/test_harness_gpu.py -q -x -k "bulk_tail or progress or tail or harness or dataset or cold" > $O/tests_new.txt 2>&1; tail -5 $O/tests_new.txt
