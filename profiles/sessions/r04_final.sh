#!/bin/bash
# round-4 final GPU session: PMC traffic of the final table (stamp), the full suite, smoke, the driver-style bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04final
mkdir -p $O
cd $R
timeout 1200 bash scripts/pmc_bench_traffic.sh gpurun_out/r04final/pmc_traffic > $O/pmc_traffic.txt 2>&1; tail -4 $O/pmc_traffic.txt
rm -f $O/pmc_traffic/*_counter_collection.csv $O/pmc_traffic/*kernel_trace.csv $O/pmc_traffic/*agent_info*
timeout 2000 python -m pytest tests -m gpu -q > $O/tests_all.txt 2>&1; tail -3 $O/tests_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; head -c 220 $O/bench.json; echo
