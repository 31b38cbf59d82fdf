#!/bin/bash
# k_extend8 knobs on the large scenes: LDS-staged nodes, group-stack levels in LDS, hand-over threshold, node stride; best of 3 device ms
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; }
for cfg in "sponza/sponza.xml 64" "disney_bsdf_test/disney_bsdf.xml 64"; do set -- $cfg
  echo "== $1 spp=$2"
  echo -n "default: "; run $1 $2
  for n in 0 9 73 200; do echo -n "lds nodes $n: "; LJ_TUNE_EXT8_NODES=$n run $1 $2; done
  for n in 4 8; do echo -n "stack levels $n: "; LJ_TUNE_EXT8_STACK=$n run $1 $2; done
  for n in 16 24 32 48 56; do echo -n "min descending $n: "; LJ_TUNE_MINDESC=$n run $1 $2; done
  for n in 4 16 24; do echo -n "refill $n: "; LJ_TUNE_REFILL=$n run $1 $2; done
  echo -n "stride 128: "; LJ_TUNE_NODE8_STRIDE=128 run $1 $2
  echo -n "stride 128, lds nodes 0: "; LJ_TUNE_NODE8_STRIDE=128 LJ_TUNE_EXT8_NODES=0 run $1 $2
done
