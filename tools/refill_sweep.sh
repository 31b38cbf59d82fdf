#!/bin/bash
# extend kernel, large scenes: refill threshold and lane count on the current defaults (best of 3 device ms)
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; }
for cfg in "sponza/sponza.xml 256" "disney_bsdf_test/disney_bsdf.xml 256"; do set -- $cfg
  echo "== $1 spp=$2"
  for rf in 2 4 8 16 24; do echo -n "refill=$rf: "; LJ_TUNE_REFILL=$rf run $1 $2; done
  for l in 1 2; do echo -n "lanes=$l: "; LJ_TUNE_LANES=$l run $1 $2; done
  for md in 40 48; do echo -n "min_descending=$md: "; LJ_TUNE_MINDESC=$md run $1 $2; done
  for p in 23 24 25 26; do echo -n "pool=2^$p: "; timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 2 0 $((1<<p)) 2>/dev/null | awk '{print $3}' | sort -n | head -1; done
done
