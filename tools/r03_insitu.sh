#!/bin/bash
# kernel averages of bench.py under rocprofv3 for name substrings: r03_insitu.sh substr...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03insitu
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python $R/bench.py --steps 10 --warmup 3 --cpu-baseline none --boundary-iters 0 --family-steps 0 --consumer-iters 0 > $OUT/run.json 2> $OUT/run.err || exit 1
cd $R
echo "== $(python -c "import json;d=json.loads(open('$OUT/run.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'])")"
for k in "$@"; do python tools/kstats.py $OUT/run "$k" | grep -v "^#"; done
