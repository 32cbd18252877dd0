#!/bin/bash
# round-4 GPU session 6: loop tail in one launch, multi-gather head, derived-bound pair outputs: tests + A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "loop_tail or pairs_out_under or sched_step or philox" > $O/tests_new.txt 2>&1; tail -5 $O/tests_new.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "blocks_golden or unet_tiny or sample_tiny or cfg1 or vae_tiny or hipgraph or full_size_properties or published_unet or cfg4 or cmdlist" > $O/tests_golden.txt 2>&1; tail -5 $O/tests_golden.txt
for round in 1 2; do
  for cfg in "1 1" "0 1" "1 0" "0 0"; do
    set -- $cfg
    MEDFUSION_LOOP_TAIL=$1 MEDFUSION_DERIVED_BOUNDS=$2 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt-path --no-other-workloads > $O/b_$1$2_$round.json 2> $O/b_$1$2_$round.err
    python - $1 $2 $round <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r04s6/b_{sys.argv[1]}{sys.argv[2]}_{sys.argv[3]}.json"))
    f = d["roofline"]["families_ms"]
    print(f"loop tail fused {sys.argv[1]} derived bounds {sys.argv[2]} round {sys.argv[3]}: {d['value']:.3f} images/s  {d['ms_per_step']:.2f} ms | {f}")
except Exception as e:
    print(sys.argv[1:], "FAILED", e)
PY
  done
done
