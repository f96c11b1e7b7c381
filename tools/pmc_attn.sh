#!/bin/bash
# SQ counters of the attention kernels at the base_det shape (run through gpurun from the repo root):
#   bash tools/pmc_attn.sh [script] -> gpurun_out/pmc_attn/*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
S=${1:-tools/bench_attn.py}
OUT=$R/gpurun_out/pmc_attn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python $R/$S > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python $R/tools/pmc_attn_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
