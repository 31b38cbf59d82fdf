#!/bin/bash
# cbox 256 spp: path-pool size x number of lanes (ms per render, best of 3)
cd $GRAFT_REPO_ROOT
for lanes in 2 3 4; do for pool in 8388608 12582912 16777216 25165824 33554432; do
  echo -n "lanes=$lanes pool=$pool: "
  LJ_TUNE_LANES=$lanes timeout -k 10 120 python3 tools/render_once.py scenes/cbox/cbox.xml 256 3 0 $pool 2>/dev/null | awk '{print $3}' | sort -n | head -1
done; done
