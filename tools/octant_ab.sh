#!/bin/bash
# survivors of a shade chunk grouped by the direction octant of their new ray (LJ_TUNE_OCTANT_BIN=1) or left in thread order (0): best-of-3 device ms
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; }
for cfg in "sponza/sponza.xml 256" "disney_bsdf_test/disney_bsdf.xml 256" "matpreview/matpreview.xml 64"; do set -- $cfg
  for v in 0 1 0 1; do echo -n "$1 @ $2 octant_bin=$v: "; LJ_TUNE_OCTANT_BIN=$v run $1 $2; done
done
for v in 0 1; do LJ_TUNE_OCTANT_BIN=$v LJ_EXTEND_STATS=1 timeout -k 10 200 python3 tools/render_once.py scenes/sponza/sponza.xml 64 1 0 2>&1 | grep "extend stats"; done
