#!/bin/bash
# cbox 256 spp: extend grid (workgroups per CU) x number of lanes x shade workgroups per CU (ms per render, best of 3)
cd $GRAFT_REPO_ROOT
for lanes in 2 3; do for ext in 3 4 5 6 8; do for sh in 8 12; do
  echo -n "lanes=$lanes ext/CU=$ext shade/CU=$sh: "
  LJ_TUNE_LANES=$lanes LJ_TUNE_EXTEND_BLOCKS_PER_CU=$ext LJ_TUNE_BLOCKS_PER_CU=$sh timeout -k 10 120 python3 tools/render_once.py scenes/cbox/cbox.xml 256 3 2>/dev/null | awk '{print $3}' | sort -n | head -1
done; done; done
