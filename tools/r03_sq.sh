#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r03
bash tools/pmc_msda_sq.sh cfg3_ext 0 > gpurun_out/r03/sq_ext_new.txt 2>&1
bash tools/pmc_msda_sq.sh cfg3_inj 0 > gpurun_out/r03/sq_inj_new.txt 2>&1
grep -A26 "msda_tile" gpurun_out/r03/sq_ext_new.txt | head -30
grep -A26 "msda_tile" gpurun_out/r03/sq_inj_new.txt | head -30
