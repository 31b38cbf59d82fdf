#!/bin/bash
# can the VALU-bound extend launches of one lane overlap the HBM-bound shade launches of another on a large tree?  lanes x extend workgroups per CU
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; }
for cfg in "sponza/sponza.xml 256" "disney_bsdf_test/disney_bsdf.xml 256"; do set -- $cfg
  echo -n "$1 @ $2 default (1 lane): "; run $1 $2
  for lanes in 2 3; do for e in 2 3 4 5; do
    echo -n "$1 lanes=$lanes extend_blocks_per_cu=$e: "; LJ_TUNE_LANES=$lanes LJ_TUNE_EXTEND_BLOCKS_PER_CU=$e run $1 $2
  done; done
done
