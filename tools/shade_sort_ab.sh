#!/bin/bash
# k_shade of scenes with several material kinds / an environment map: the whole segment sorted by what a path will do (default, LJ_TUNE_SHADE_SORT=2)
# against every 256-path chunk sorted on its own (1).  best-of-3 device ms, then per-kernel timing mode
cd $GRAFT_REPO_ROOT
for cfg in "disney_bsdf_test/disney_bsdf.xml 256" "disney_bsdf_test/disney_bsdf.xml 64" "matpreview/matpreview.xml 64"; do set -- $cfg
  [ -f scenes/$1 ] || continue
  for v in 2 1; do
    echo "== $1 spp=$2 sort=$v"
    LJ_TUNE_SHADE_SORT=$v timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1
    LJ_TUNE_SHADE_SORT=$v timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 1 1 2>/dev/null | tail -1
  done
done
