#!/bin/bash
# PMC passes over one conv shape (each --pmc set in its own run, as the MI355X guide prescribes).
# usage: scripts/pmc_conv.sh <outdir> <tile> <splitk> [shape] [sets: all|core] [precision]
OUT=$1; TILE=$2; SK=$3; SHAPE=${4:-16,32,32,256,0,256,3,1,0}; WHICH=${5:-all}; PREC=${6:-0}
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
SETS=("GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS")
if [ "$WHICH" = "all" ]; then
  SETS+=("SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum")
fi
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d $R/$OUT -o pmc$i --output-format csv -- python $R/scripts/conv_one.py --shape $SHAPE --tile $TILE --splitk $SK --precision $PREC --reps 10 > $R/$OUT/run$i.log 2>&1
done
KN=conv_igemm; [ "$PREC" = "5" ] && KN=conv_f16x2
python $R/scripts/pmc_summary.py $R/$OUT $KN
