#!/bin/bash
# extend kernel: BVH8 (k_extend8) against the BVH4 kernel on the trees beyond the LDS image; best of 3 device ms + the kernel's own step counters
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; }
for cfg in "sponza/sponza.xml 64" "sponza/sponza.xml 256" "disney_bsdf_test/disney_bsdf.xml 64" "disney_bsdf_test/disney_bsdf.xml 256"; do set -- $cfg
  echo "== $1 spp=$2"
  for v in 1 0; do
    echo -n "bvh8=$v: "; LJ_TUNE_BVH8=$v run $1 $2
  done
done
for sc in sponza/sponza.xml disney_bsdf_test/disney_bsdf.xml; do
  for v in 1 0; do
    LJ_TUNE_BVH8=$v LJ_EXTEND_STATS=1 timeout -k 10 200 python3 tools/render_once.py scenes/$sc 64 1 0 2>&1 | grep "extend stats"
  done
done
LJ_TUNE_BVH8=1 timeout -k 10 200 python3 tools/render_once.py scenes/sponza/sponza.xml 64 2 1 2>&1 | tail -1
LJ_TUNE_BVH8=0 timeout -k 10 200 python3 tools/render_once.py scenes/sponza/sponza.xml 64 2 1 2>&1 | tail -1
