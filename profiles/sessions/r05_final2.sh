#!/bin/bash
# round-5 closing session: the full GPU suite, smoke and the driver-style bench line on the final tree (the PMC / rocprofv3 artefacts of r05_final.sh
# stay valid: no convolution source changed since)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05final2
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/tests_all.txt 2>&1; tail -4 $O/tests_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
