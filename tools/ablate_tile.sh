#!/bin/bash
# timing-only ablation of the tile kernel (results are wrong when a bit is set): see VAH_TILE_ABLATE in csrc/msda_tile.hip
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for a in ${ABL:-0 1 2 4 16 23 64}; do
  for c in cfg3_ext cfg3_inj; do
    VAH_TILE_ABLATE=$a timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl/${c}_$a -- python $R/tools/prof_msda_single.py $c 4 0 > /dev/null 2>&1
    echo -n "ablate=$a $c: "; python $R/tools/kstats.py $R/gpurun_out/abl/${c}_$a msda_tile | grep -o "avg *[0-9.]* us"
  done
done
