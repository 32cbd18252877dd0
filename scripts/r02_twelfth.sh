#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02l_prof
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02l_prof -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-path --no-roofline > $R/gpurun_out/r02l_prof_bench.json 2> $R/gpurun_out/r02l_prof_bench.err
ls -la $R/gpurun_out/r02l_prof
python $R/scripts/trace_gaps.py $R/gpurun_out/r02l_prof/bench_kernel_trace.csv $R/gpurun_out/r02l_trace_gaps.txt | head -40
cp $R/gpurun_out/r02l_prof/bench_kernel_stats.csv $R/gpurun_out/r02l_kernel_stats.csv
rm -f $R/gpurun_out/r02l_prof/bench_kernel_trace.csv
cd $R
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path --graph > gpurun_out/r02l_bench_graph.json 2> gpurun_out/r02l_bench_graph.err; tail -c 600 gpurun_out/r02l_bench_graph.json
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-path > gpurun_out/r02l_bench_eager.json 2> gpurun_out/r02l_bench_eager.err; head -c 300 gpurun_out/r02l_bench_eager.json
timeout 1500 bash scripts/pmc_bench_traffic.sh gpurun_out/r02l_pmc_traffic
rm -f gpurun_out/r02l_pmc_traffic/*_counter_collection.csv gpurun_out/r02l_pmc_traffic/*kernel_trace.csv
