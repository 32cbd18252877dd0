#!/bin/bash
# round-5 GPU session 9: the full GPU suite, smoke, the driver-style bench line with the Winograd table in the library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s9
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests_all.txt 2>&1; tail -5 $O/tests_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo; tail -3 $O/bench.err
