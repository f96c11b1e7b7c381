#!/bin/bash
# A/B of one environment switch on the bench under rocprofv3: r03_ab.sh VAR [kernel-name substring ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03ab
rm -rf $OUT; mkdir -p $OUT
VAR=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/new -- python $R/bench.py --steps 10 --warmup 3 --cpu-baseline none --boundary-iters 0 --family-steps 0 > $OUT/new.json 2> $OUT/new.err || exit 1
export $VAR=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/old -- python $R/bench.py --steps 10 --warmup 3 --cpu-baseline none --boundary-iters 0 --family-steps 0 > $OUT/old.json 2> $OUT/old.err || exit 1
cd $R
for v in new old; do
  echo "== $v: $(python -c "import json;d=json.loads(open('$OUT/$v.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'])")"
  for k in "$@"; do python tools/kstats.py $OUT/$v "$k" | grep -v "^#"; done
done
