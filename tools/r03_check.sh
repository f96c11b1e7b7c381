#!/bin/bash
# Round-3 acceptance loop on the GPU box: the whole -m gpu suite, then the driver-style bench line (gpurun_out/r03/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r03
cd $R
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r03/pytest_gpu.log 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/r03/pytest_gpu.log
VAH_GEMM_TABLE_DUMP=gpurun_out/r03/gemm_table.txt python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['config'].get('gemm_candidates_rejected'))
for k,v in d['kernels'].items():
    if k.startswith(('gemm','msda_fused')): print(k, v.get('avg_us'), v.get('frac_of_hbm_peak', v.get('frac_of_mfma_peak')))
PY
