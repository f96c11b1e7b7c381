#!/bin/bash
# parity tests of the MSDA kernels + micro-benchmarks (run through gpurun from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r03
cd $R
timeout -k 10 900 python -m pytest tests/test_msda_gpu.py tests/test_msda_fullsize_fused_gpu.py "tests/test_backbone_gpu.py::test_fused_backward_tile_pass_equals_atomics" tests/test_mmcv_attention.py -x -q -m gpu > gpurun_out/r03/msda_tests.log 2>&1
echo "tests rc=$?"
tail -5 gpurun_out/r03/msda_tests.log
timeout -k 10 300 python tools/bench_msda_fused.py > gpurun_out/r03/msda_fused_ubench.txt 2>&1
cat gpurun_out/r03/msda_fused_ubench.txt
cd /tmp && export TMPDIR=/tmp
for c in cfg3_ext cfg3_inj; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/ks_$c -- python $R/tools/prof_msda_single.py $c 5 0 > /dev/null 2>&1
  python $R/tools/kstats.py $R/gpurun_out/r03/ks_$c msda
done
