#!/bin/bash
# Single-GPU bench lines of the other BASELINE configurations (configs[1], [3], [4]; [2] is bench.py's default)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03presets
mkdir -p $OUT
cd $R
run() {  # name, args...
  n=$1; shift
  timeout -k 10 500 python bench.py --steps 10 --warmup 3 --cpu-baseline none --boundary-iters 0 --family-steps 0 --consumer-iters 0 "$@" > $OUT/$n.json 2> $OUT/$n.err
  echo "$n rc=$? $(python -c "
import json
try:
    d=json.loads(open('$OUT/$n.json').read().strip().splitlines()[-1]); print(d['value'], 'img/s', d['ms_per_step'], 'ms', 'graph' if d['config']['hip_graph'] else 'eager', d['config'].get('hip_graph_probe'))
except Exception as e: print('no line', e)")"
}
run tiny_seg_512 --preset tiny_seg --size 512 512 --batch 2
run large_seg_640 --preset large_seg --size 640 640 --batch 2
run large_seg_640_nocp --preset large_seg --size 640 640 --batch 2 --no-checkpoint
run beit_large_seg_640 --preset beit_large_seg --size 640 640 --batch 2
run large_seg_800x1344 --preset large_seg --size 800 1344 --batch 1
run large_seg_800x1344_nocp --preset large_seg --size 800 1344 --batch 1 --no-checkpoint
