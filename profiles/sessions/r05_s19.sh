#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s19
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -k "pairs_only or golden or trained_like or cfg2_rows or full_size" > $O/tests.txt 2>&1; tail -4 $O/tests.txt
for v in 0 1; do
  MEDFUSION_PAIRS_ONLY_OUTPUTS=$v timeout 600 python scripts/wino_ab.py $O/ab_po$v.json 2 > $O/ab_po$v.txt 2>&1; echo "pairs-only block outputs=$v: $(grep 'winograd  ' $O/ab_po$v.txt)"
done
