#!/bin/bash
# the two bench workloads with trees beyond the LDS image at their BASELINE sample counts: BVH8 (k_extend8) against the BVH4 extend kernel
cd $GRAFT_REPO_ROOT
for cfg in "sponza/sponza.xml 1024" "disney_bsdf_test/disney_bsdf.xml 256"; do set -- $cfg
  for v in 1 0; do
    echo "== $1 spp=$2 bvh8=$v"
    LJ_TUNE_BVH8=$v timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 2 0 2>/dev/null | tail -1
    LJ_TUNE_BVH8=$v timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 1 1 2>/dev/null | tail -1
  done
done
