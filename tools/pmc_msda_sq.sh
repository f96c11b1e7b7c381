#!/bin/bash
# SQ counters of the MSDA kernels at a BASELINE call shape (run through gpurun from the repo root):
#   bash tools/pmc_msda_sq.sh cfg3_ext [noise] -> gpurun_out/pmc_msda_<cfg>/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=${1:-cfg3_ext}
NOISE=${2:-0}
OUT=$R/gpurun_out/pmc_msda_$C
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $R/tools/prof_msda_single.py $C 3 $NOISE > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python $R/tools/pmc_summary.py $OUT msda > $OUT/summary.txt
cat $OUT/summary.txt
