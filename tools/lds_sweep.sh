#!/bin/bash
# extend kernel, large scenes: LDS image size x stack levels (workgroups per CU follow from the LDS a workgroup takes: image + 9 KiB of leaf pools)
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python3 tools/render_once.py scenes/$1 64 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; }
for sc in sponza/sponza.xml disney_bsdf_test/disney_bsdf.xml; do
  echo "== $sc"
  for kb in 14 16 18 20 22 24; do for st in 8 12; do echo -n "ext_lds_kb=$kb stack=$st: "; LJ_TUNE_EXT_LDS_KB=$kb LJ_TUNE_EXT_STACK=$st run $sc; done; done
done
