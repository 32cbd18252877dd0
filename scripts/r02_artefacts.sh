#!/bin/bash
# round-2 artefact session: rocprofv3 kernel stats + gap analysis of the bench command, PMC passes of the conv kernel (two tiles), HBM traffic of
# the bench command (PMC), the bench line itself, the full GPU suite
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02f_prof
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02f_prof -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-path --no-roofline > $R/gpurun_out/r02f_prof_bench.json 2> $R/gpurun_out/r02f_prof_bench.err
python $R/scripts/trace_gaps.py $R/gpurun_out/r02f_prof/bench_kernel_trace.csv $R/gpurun_out/r02f_trace_gaps.txt | head -12
cp $R/gpurun_out/r02f_prof/bench_kernel_stats.csv $R/gpurun_out/r02f_kernel_stats.csv
rm -f $R/gpurun_out/r02f_prof/bench_kernel_trace.csv
cd $R
timeout 900 bash scripts/pmc_conv.sh gpurun_out/r02f_pmc_t33 33 1 16,32,32,256,0,256,3,1,0 all 5 > gpurun_out/r02f_pmc_t33.txt 2>&1
timeout 900 bash scripts/pmc_conv.sh gpurun_out/r02f_pmc_t31 31 8 16,8,8,1024,1024,1024,3,1,0 all 5 > gpurun_out/r02f_pmc_t31.txt 2>&1
rm -f gpurun_out/r02f_pmc_t3*/*_counter_collection.csv gpurun_out/r02f_pmc_t3*/*kernel_trace.csv
timeout 2400 bash scripts/pmc_bench_traffic.sh gpurun_out/r02f_pmc_traffic > gpurun_out/r02f_pmc_traffic.txt 2>&1; tail -8 gpurun_out/r02f_pmc_traffic.txt
rm -f gpurun_out/r02f_pmc_traffic/*_counter_collection.csv gpurun_out/r02f_pmc_traffic/*kernel_trace.csv
cp gpurun_out/r02f_pmc_traffic/traffic.json profiles/pmc_bench_traffic.json
timeout 1200 python bench.py --steps 3 --warmup 1 > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err; head -c 400 gpurun_out/r02f_bench.json
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r02f_t_all.log 2>&1; tail -4 gpurun_out/r02f_t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02f_smoke.log 2>&1; tail -2 gpurun_out/r02f_smoke.log
