#!/bin/bash
# run-time knobs of the wavefront plan on the large scenes (best of 2 device ms, 64 spp)
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python3 tools/render_once.py scenes/$1 64 2 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; }
for sc in sponza/sponza.xml disney_bsdf_test/disney_bsdf.xml; do
  echo "== $sc"
  for ml in 2 3 4 6 8; do echo -n "max_leaf=$ml: "; LJ_TUNE_MAX_LEAF=$ml run $sc; done
  for md in 8 16 24 32 40; do echo -n "min_descending=$md: "; LJ_TUNE_MINDESC=$md run $sc; done
  for rf in 4 8 16 24; do echo -n "refill=$rf: "; LJ_TUNE_REFILL=$rf run $sc; done
  for kb in 16 24 40 64; do echo -n "ext_lds_kb=$kb: "; LJ_TUNE_EXT_LDS_KB=$kb run $sc; done
  for st in 8 12 16 20; do echo -n "ext_stack=$st: "; LJ_TUNE_EXT_STACK=$st run $sc; done
  for ex in 2 4 8 12; do echo -n "extend_blocks_per_cu=$ex (lanes=1): "; LJ_TUNE_LANES=1 LJ_TUNE_EXTEND_BLOCKS_PER_CU=$ex run $sc; done
done
