#!/bin/bash
cd $GRAFT_REPO_ROOT
for lanes in 4 6 8; do for ext in 1 2; do for pool in 33554432 50331648 67108864; do
  echo -n "lanes=$lanes ext/CU=$ext pool=$pool: "
  LJ_TUNE_LANES=$lanes LJ_TUNE_EXTEND_BLOCKS_PER_CU=$ext timeout -k 10 120 python3 tools/render_once.py scenes/cbox/cbox.xml 256 3 0 $pool 2>/dev/null | awk '{print $3}' | sort -n | head -1
done; done; done
