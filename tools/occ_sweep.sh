#!/bin/bash
# usage (on the GPU box): tools/occ_sweep.sh — (variants are built into liblajolla_hip_occ.so, never into the default library) A/B of compile-time occupancy targets under the two-lane render
for flag in "-DLJ_LAMBERT_OCC=4" "-DLJ_LAMBERT_OCC=5" "-DLJ_LAMBERT_OCC=6" "-DLJ_LAMBERT_OCC=6 -DLJ_EXT_RESIDENT_OCC=6"; do
  touch lajolla_public_amd/csrc/device/kernels.hip
  LJ_VARIANT=occ LJ_EXTRA_HIPCC_FLAGS="$flag" python -m lajolla_public_amd.build > /dev/null 2>&1
  echo -n "[$flag]: "; LJ_VARIANT=occ timeout -k 10 120 python tools/render_once.py scenes/cbox/cbox.xml 256 3 0 2>&1 | tail -1
done
