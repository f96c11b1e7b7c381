#!/bin/bash
# MSDA parity tests + kernel durations and FETCH_SIZE of the fused call shapes (quick loop while tuning the MSDA kernels)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03quick
rm -rf $OUT; mkdir -p $OUT
cd $R
if [ "${1:-tests}" = tests ]; then
  timeout -k 10 600 python -m pytest tests/test_msda_gpu.py tests/test_msda_fullsize_fused_gpu.py tests/test_backbone_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1
  echo "tests rc=$?"; tail -3 $OUT/pytest.log
fi
cd /tmp && export TMPDIR=/tmp
for c in cfg3_inj cfg3_ext; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats/fused_$c -- python $R/tools/prof_msda_single.py $c 5 0 pair > $OUT/stats_fused_$c.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/${c}_n0_FETCH_SIZE -- python $R/tools/prof_msda_single.py $c 3 0 pair > $OUT/pmc_$c.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc/${c}_n0_RDREQ -- python $R/tools/prof_msda_single.py $c 3 0 pair > $OUT/pmc_rd_$c.log 2>&1 || echo "rdreq pass failed: $(tail -2 $OUT/pmc_rd_$c.log)"
done
cd $R
python tools/kstats.py $OUT/stats msda
python - <<'PY'
import csv,glob,os,re
from collections import defaultdict
root=os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r03quick/pmc')
for d in sorted(glob.glob(root+'/*SIZE')+glob.glob(root+'/*RDREQ')):
    fs=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not fs: continue
    acc=defaultdict(lambda: defaultdict(float)); cnt=defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(fs[0])):
        n=re.sub(r'\(.*$','',re.sub(r'\(anonymous namespace\)::|vah::|void ','',r['Kernel_Name']))[:40]
        if 'msda' in n: acc[n][r['Counter_Name']]+=float(r['Counter_Value']); cnt[n][r['Counter_Name']]+=1
    print(os.path.basename(d))
    for n in acc:
        v={c: acc[n][c]/cnt[n][c] for c in acc[n]}
        if 'FETCH_SIZE' in v: print('   %-42s %8.1f MiB/launch'%(n, v['FETCH_SIZE']/1024))
        else:
            n32,n64,n128=(v.get('TCC_EA0_RDREQ_%s_sum'%k,0) for k in ('32B','64B','128B'))
            tot=v.get('TCC_EA0_RDREQ_sum',0)
            print('   %-42s req %9.0f (32B %9.0f 64B %9.0f 128B %9.0f) -> %7.1f MiB by size'%(n,tot,n32,n64,n128,(32*n32+64*n64+128*n128)/2**20))
PY
