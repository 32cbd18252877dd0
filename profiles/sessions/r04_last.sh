#!/bin/bash
# round-4 last GPU session: the whole GPU suite on the final tree, smoke, a short bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04last
mkdir -p $O
cd $R
timeout 420 python -m pytest tests -m gpu -q -x > $O/tests_all.txt 2>&1; tail -3 $O/tests_all.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 150 python bench.py --gpus 1 --steps 5 --warmup 2 --no-other-workloads > $O/bench.json 2> $O/bench.err; head -c 200 $O/bench.json; echo
