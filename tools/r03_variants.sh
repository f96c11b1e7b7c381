#!/bin/bash
# A/B of compile-time variants of the tile pass on one box: rebuilds msda_tile.o with -D flags, relinks, runs the fused microbench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r03
OBJS=$(ls vit-adapter_amd/build/*.o | grep -v msda_tile.o)
for v in "$@"; do
  flags=$(echo $v | tr ',' ' ')
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=fast $flags -c vit-adapter_amd/csrc/msda_tile.hip -o /tmp/msda_tile_v.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vit-adapter_amd/lib/libvitadapter_hip.so $OBJS /tmp/msda_tile_v.o -L/opt/rocm/lib -lhipblaslt || exit 1
  echo "== variant: $v"
  timeout -k 10 200 python tools/bench_msda_fused.py 2>&1 | grep "tiled=1"
  (cd /tmp && export TMPDIR=/tmp && for c in cfg3_ext cfg3_inj; do timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/var_$c -- python $R/tools/prof_msda_single.py $c 5 0 > /dev/null 2>&1; python $R/tools/kstats.py $R/gpurun_out/r03/var_$c msda_tile | grep -o "avg *[0-9.]* us"; rm -rf $R/gpurun_out/r03/var_$c; done)
done
