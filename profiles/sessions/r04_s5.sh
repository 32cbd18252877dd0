#!/bin/bash
# round-4 GPU session 5: write-through (sc1) output stores of the convolution / the apply pass: A/B by library twin, two interleaved rounds
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s5
mkdir -p $O
cd $R
V=$R/medfusion_amd/csrc/build/variants
for round in 1 2; do
  for lib in default sc1conv sc1gn sc1both; do
    if [ $lib = default ]; then unset MEDFUSION_LIB; else export MEDFUSION_LIB=$V/libmedfusion_hip_$lib.so; fi
    timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt-path --no-other-workloads > $O/b_${lib}_$round.json 2> $O/b_${lib}_$round.err
    python - $lib $round <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r04s5/b_{sys.argv[1]}_{sys.argv[2]}.json"))
    f = d["roofline"]["families_ms"]
    print(f"{sys.argv[1]:8s} round {sys.argv[2]}: {d['value']:.3f} images/s  {d['ms_per_step']:.2f} ms | conv {f.get('conv_igemm')} gn_apply {f.get('gn_apply')}")
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e)
PY
  done
done
unset MEDFUSION_LIB
