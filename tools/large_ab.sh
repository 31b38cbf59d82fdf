#!/bin/bash
# large scenes: SBVH on / off (LJ_TUNE_SBVH) x lanes, best of 3 device ms
cd $GRAFT_REPO_ROOT
for s in "disney_bsdf_test/disney_bsdf.xml 64" "sponza/sponza.xml 64"; do set -- $s
  for sb in 0 1; do for lanes in 1 2; do echo -n "$1 spp=$2 sbvh=$sb lanes=$lanes: "; LJ_TUNE_SBVH=$sb LJ_TUNE_LANES=$lanes timeout -k 10 200 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; done; done
  LJ_EXTEND_STATS=1 timeout -k 10 200 python3 tools/render_once.py scenes/$1 $2 1 0 2>&1 | grep "extend stats"
done
