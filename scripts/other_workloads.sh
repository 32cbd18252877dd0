#!/bin/bash
# the other BASELINE configs at their per-GPU size (and the reduced-precision opt-in modes), one bench line each -> $1 (default gpurun_out/r03_other_workloads.txt)
cd /root/repo; mkdir -p gpurun_out
OUT=${1:-gpurun_out/r03_other_workloads.txt}
for wl in cfg3_g8 cfg3_g1 cfg5 cfg4; do
  st=2; [ $wl = cfg4 ] && st=1
  timeout 900 python bench.py --workload $wl --steps $st --warmup 1 --no-cpu-baseline --no-roofline --alt-precision 1 2> gpurun_out/ow_$wl.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$wl', d['value'], 'images/s', d['ms_per_step'], 'ms/step |', d['config']['workload'], '| exact bf16 triplets:', [a['value'] for a in d.get('other_conv_arithmetic', [])])"
done | tee $OUT
timeout 900 python bench.py --conv-precision 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-alt-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('cfg2 opt-in reduced precision (MF_CONV_BF16)', d['value'], 'images/s')" | tee -a $OUT
timeout 900 python bench.py --conv-precision 6 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-alt-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('cfg2 opt-in reduced precision (MF_CONV_F16: one fp16 term on the LDS-DMA kernel)', d['value'], 'images/s')" | tee -a $OUT
timeout 900 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-alt-path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('cfg2 at 32 images per GPU', d['value'], 'images/s')" | tee -a $OUT
