#!/bin/bash
set -x
mkdir -p gpurun_out
cd /root/repo
for z in "" "--zeros"; do
  for t in "31 8 16,8,8,1024,1024,1024,3,1,0" "33 1 16,32,32,256,0,256,3,1,0" "51 8 16,8,8,1024,1024,1024,3,1,0"; do
    set -- $t
    timeout 120 python scripts/conv_one.py --precision 5 --tile $1 --splitk $2 --shape $3 --reps 30 $z 2>&1 | tail -1
  done
done > gpurun_out/r02f_zeros.txt 2>&1
cat gpurun_out/r02f_zeros.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02f_prof -o p5 --output-format csv -- python /root/repo/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-alt-path --no-roofline --conv-precision 5 > /root/repo/gpurun_out/r02f_prof.log 2>&1
cd /root/repo
ls gpurun_out/r02f_prof | head; head -25 gpurun_out/r02f_prof/p5_kernel_stats.csv | cut -c1-220
timeout 600 python bench.py --steps 2 --warmup 1 --conv-precision 5 --no-cpu-baseline --alt-precision 1 > gpurun_out/r02f_bench_p5.json 2> gpurun_out/r02f_bench_p5.err; cut -c1-300 gpurun_out/r02f_bench_p5.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02f_bench_p5.json'))
print(d['value'], d['roofline']['families_ms'], d['other_conv_arithmetic'])
PY
