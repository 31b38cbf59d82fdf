#!/bin/bash
# wavefront plan: paths in flight (queue = 128 B per path) at the bench configurations (best of 2 device ms)
cd $GRAFT_REPO_ROOT
for cfg in "sponza/sponza.xml 1024" "disney_bsdf_test/disney_bsdf.xml 256"; do set -- $cfg
  echo "== $1 spp=$2"
  for p in 25 26 27 28; do echo -n "pool=2^$p: "; timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 2 0 $((1<<p)) 2>/dev/null | awk '{print $3}' | sort -n | head -1; done
done
