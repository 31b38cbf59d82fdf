#!/bin/bash
# host-side BVH build knobs against device time: leaf size cap x SAH primitive cost (large scenes, 64 spp, best of 3 device ms)
cd $GRAFT_REPO_ROOT
for s in "disney_bsdf_test/disney_bsdf.xml 64" "sponza/sponza.xml 64"; do set -- $s
  for ml in 2 4 8; do for pc in 0.6 1.2 2.4; do echo -n "$1 spp=$2 max_leaf=$ml prim_cost=$pc: "; LJ_TUNE_MAX_LEAF=$ml LJ_TUNE_PRIM_COST=$pc timeout -k 10 200 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; done; done
done
