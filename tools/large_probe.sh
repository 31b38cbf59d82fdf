#!/bin/bash
# large scenes: lanes sweep + extend-kernel utilisation counters (developer probe)
cd $GRAFT_REPO_ROOT
for s in "disney_bsdf_test/disney_bsdf.xml 64" "sponza/sponza.xml 64"; do set -- $s
  for lanes in 1 2 4; do echo -n "$1 spp=$2 lanes=$lanes: "; LJ_TUNE_LANES=$lanes timeout -k 10 200 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; done
  LJ_EXTEND_STATS=1 timeout -k 10 200 python3 tools/render_once.py scenes/$1 $2 1 0 2>&1 | grep "extend stats"
done
