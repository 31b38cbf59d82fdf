#!/bin/bash
# usage (on the GPU box): tools/occ_sweep.sh — rebuilds the kernels with different occupancy targets for the resident extend variant
for occ in 4 5 6 8; do
  touch lajolla_public_amd/csrc/device/kernels.hip
  LJ_EXTRA_HIPCC_FLAGS="-DLJ_EXT_RESIDENT_OCC=$occ" python -m lajolla_public_amd.build > /dev/null 2>&1
  echo -n "resident extend occ=$occ: "; timeout -k 10 120 python tools/render_once.py scenes/cbox/cbox.xml 256 3 1 2>&1 | tail -1
done
