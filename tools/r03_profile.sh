#!/bin/bash
# Round-3 profile collection on the GPU box (run through gpurun from the repo root):
#   bash tools/r03_profile.sh
# kernel-trace stats of bench.py and of the MSDA call shapes (the call form of the module: fp32 pair rows), and the
# FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, kernel-trace only) that profiles/r03_msda_pmc.json is made from.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in cfg3_inj cfg3_ext; do
  for n in 0 1; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc/${c}_n${n}_${ctr} -- python $R/tools/prof_msda_single.py $c 3 $n pair > $OUT/pmc_${c}_${n}_${ctr}.log 2>&1 || exit 1
    done
  done
  echo "pmc $c done"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats/fused_$c -- python $R/tools/prof_msda_single.py $c 5 0 pair > $OUT/stats_fused_$c.log 2>&1 || exit 1
done
for c in cfg3_inj cfg3_ext cfg1 cfg5_pixdec; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats/plain_$c -- python $R/tools/prof_msda_plain.py $c 5 > $OUT/stats_plain_$c.log 2>&1 || exit 1
done
echo "stats done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats/bench -- python $R/bench.py --steps 10 --warmup 3 --cpu-baseline none --boundary-iters 0 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || exit 1
cd $R
python tools/pmc_msda_summary.py $OUT/pmc 3 > $OUT/r03_msda_pmc.json
python tools/kstats.py $OUT/stats msda > $OUT/r03_msda_kernel_stats.txt
cp $(find $OUT/stats/bench -name '*kernel_stats.csv' | head -1) $OUT/r03_bench_kernel_stats.csv
echo profile-done
