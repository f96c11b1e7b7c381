#!/bin/bash
# Round-2 profile collection on the GPU box (run through gpurun from the repo root):
#   bash tools/r02_profile.sh
# kernel-trace stats of bench.py and of the MSDA call shapes, and the FETCH_SIZE / WRITE_SIZE passes
# (separate --pmc runs, kernel-trace only) that profiles/r02_msda_pmc.json is made from.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in cfg3_inj cfg3_ext; do
  for n in 0 1; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc/${c}_n${n}_${ctr} -- python $R/tools/prof_msda_single.py $c 3 $n > $OUT/pmc_${c}_${n}_${ctr}.log 2>&1 || exit 1
    done
  done
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats/fused_$c -- python $R/tools/prof_msda_single.py $c 5 0 > $OUT/stats_fused_$c.log 2>&1 || exit 1
done
for c in cfg3_inj cfg3_ext cfg1 cfg5_pixdec; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats/plain_$c -- python $R/tools/prof_msda_plain.py $c 5 > $OUT/stats_plain_$c.log 2>&1 || exit 1
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats/bench -- python $R/bench.py --steps 10 --warmup 3 --cpu-baseline none --boundary-iters 0 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || exit 1
cd $R
python tools/pmc_msda_summary.py $OUT/pmc 3 > $OUT/r02_msda_pmc.json
python tools/kstats.py $OUT/stats msda > $OUT/r02_msda_kernel_stats.txt
python tools/bench_msda.py > $OUT/r02_msda_microbench.txt 2>&1
python tools/bench_msda_fused.py > $OUT/r02_msda_fused_microbench.txt 2>&1
echo profile-done
