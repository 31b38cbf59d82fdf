#!/bin/bash
# usage (on the GPU box): tools/pool_sweep.sh  — queue pool size x workgroups per CU
for scene in "cbox/cbox.xml 256" "sponza/sponza.xml 32"; do
  set -- $scene
  for bpc in 8 12 16 32; do for pool in 16777216 33554432; do
    echo -n "$1 spp=$2 blocks/cu=$bpc pool=$pool: "
    LJ_TUNE_BLOCKS_PER_CU=$bpc timeout -k 10 120 python tools/render_once.py scenes/$1 $2 2 0 $pool 2>&1 | tail -1 | cut -d' ' -f3-7
  done; done
done
