#!/bin/bash
# bench.py with a ONE-rank RCCL process group (DDP + SyncBatchNorm collectives, captured) under rocprofv3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03onerank
rm -rf $OUT; mkdir -p $OUT
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 VAH_ONE_RANK_GROUP=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python $R/bench.py --steps 10 --warmup 3 --cpu-baseline none --boundary-iters 0 --family-steps 0 --consumer-iters 0 "$@" > $OUT/run.json 2> $OUT/run.err || { tail -5 $OUT/run.err; exit 1; }
cd $R
python - <<PY
import csv,glob,json,re
d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['config']['hip_graph'], d['config']['hip_graph_probe'])
f=glob.glob('$OUT/run/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs'])):
    n=r['Name']
    if 'nccl' in n.lower() or 'rccl' in n.lower() or 'AllReduce' in n or 'Broadcast' in n or 'copyBuffer' in n or 'fillBuffer' in n or 'cat' in n.lower() or 'foreach' in n.lower() or 'multi_tensor' in n:
        print('%6s calls avg %8.1f us total %8.2f ms  %s'%(r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, n[:120]))
PY
