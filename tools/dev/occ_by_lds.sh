#!/bin/bash
# How the BVH4 extend kernel's time follows its occupancy, with the kernel and its LDS tree image unchanged: unused stack levels in LDS
# take a workgroup from 30 KiB (five per CU) to 38 (four) and 50 (three).  Best-of-3 device ms per render and the extend launches alone.
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 3 1 2>/dev/null | sort -n -k3 | head -1 | awk '{print $3, "ms; extend", $(NF-4), "ms, shade", $(NF-1), "ms"}'; }
for cfg in "sponza/sponza.xml 256" "disney_bsdf_test/disney_bsdf.xml 256"; do set -- $cfg
  echo -n "$1 @ $2, five workgroups per CU (default): "; run $1 $2
  echo -n "$1 @ $2, four: "; LJ_TUNE_EXT_LDS_KB=28 LJ_TUNE_EXT_STACK=20 run $1 $2
  echo -n "$1 @ $2, three: "; LJ_TUNE_EXT_LDS_KB=40 LJ_TUNE_EXT_STACK=32 run $1 $2
done
