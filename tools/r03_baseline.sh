#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r03
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_base.json 2> gpurun_out/r03/bench_base.err
echo bench-done
bash tools/pmc_msda_sq.sh cfg3_ext 0 > gpurun_out/r03/sq_ext.txt 2>&1
bash tools/pmc_msda_sq.sh cfg3_inj 0 > gpurun_out/r03/sq_inj.txt 2>&1
echo sq-done
bash tools/ablate_tile.sh > gpurun_out/r03/ablate.txt 2>&1
echo ablate-done
