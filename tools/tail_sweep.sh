#!/bin/bash
# wavefront plan, large scenes: the fused tail launch on / off / its threshold, per lane count (best of 2 device ms)
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 2 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; }
for cfg in "sponza/sponza.xml 64" "sponza/sponza.xml 256" "disney_bsdf_test/disney_bsdf.xml 64" "disney_bsdf_test/disney_bsdf.xml 256"; do set -- $cfg
  echo "== $1 spp=$2"
  for lanes in 1 4; do
    echo -n "lanes=$lanes tail=0: "; LJ_TUNE_EXTEND_BLOCKS_PER_CU=5 LJ_TUNE_MINDESC=40 LJ_TUNE_LANES=$lanes LJ_TUNE_TAIL=0 run $1 $2
    for f in 1 2 4 8; do echo -n "lanes=$lanes tail_frac=$f/16: "; LJ_TUNE_EXTEND_BLOCKS_PER_CU=5 LJ_TUNE_MINDESC=40 LJ_TUNE_LANES=$lanes LJ_TUNE_TAIL_FRAC=$f run $1 $2; done
  done
done
