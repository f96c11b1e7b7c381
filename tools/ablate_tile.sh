#!/bin/bash
# Timing-only ablation of the tile kernel: builds the library with -DVAH_TILE_ABLATE=<bits> (see csrc/msda_tile.hip; the
# results of such a build are WRONG), times the kernel under rocprofv3, and rebuilds the real library at the end.
# Run through gpurun from the repo root; the rebuilt library only exists on the GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for a in ${ABL:-0 1 2 4 16 23 64}; do
  (cd $R && touch vit-adapter_amd/csrc/msda_tile.hip && VAH_EXTRA_HIPCC_FLAGS="-DVAH_TILE_ABLATE=$a" python vit-adapter_amd/build.py > /dev/null) || exit 1
  for c in cfg3_ext cfg3_inj; do
    (cd /tmp && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl/${c}_$a -- python $R/tools/prof_msda_single.py $c 4 0 > /dev/null 2>&1)
    echo -n "ablate=$a $c: "; python $R/tools/kstats.py $R/gpurun_out/abl/${c}_$a msda_tile | grep -o "avg *[0-9.]* us"
  done
done
cd $R && touch vit-adapter_amd/csrc/msda_tile.hip && python vit-adapter_amd/build.py > /dev/null
