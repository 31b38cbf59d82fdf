#!/bin/bash
# usage (on the GPU box): tools/pool_sweep2.sh — small path pools (queue records resident in the 256 MiB Infinity Cache?) x lanes
cd $GRAFT_REPO_ROOT
for pool in 524288 1048576 2097152 4194304 8388608 33554432; do for lanes in 1 2 4; do
  echo -n "cbox 256spp pool=$pool lanes=$lanes: "
  LJ_TUNE_LANES=$lanes timeout -k 10 120 python3 tools/render_once.py scenes/cbox/cbox.xml 256 3 0 $pool 2>/dev/null | awk '{print $3}' | sort -n | head -1
done; done
