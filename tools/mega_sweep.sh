#!/bin/bash
# usage (on the GPU box): tools/mega_sweep.sh — run-time knobs of the fused kernel (resident workgroups per CU x samples per grab), cbox 256 spp
cd $GRAFT_REPO_ROOT
for per_cu in 2 3 4 5 6 8; do for grab in 256 1024 4096; do
  echo -n "cbox 256spp blocks/cu=$per_cu grab=$grab: "
  LJ_TUNE_MEGA_BLOCKS_PER_CU=$per_cu LJ_TUNE_MEGA_GRAB=$grab timeout -k 10 120 python3 tools/render_once.py scenes/cbox/cbox.xml 256 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1
done; done
