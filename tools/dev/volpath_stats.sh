#!/bin/bash
# where k_volpath's instructions go: wave-level event counts and lane occupancy per part of the tracer.  Needs the instrumented variant:
#   LJ_VARIANT=vstats LJ_EXTRA_HIPCC_FLAGS="-DLJ_VOLPATH_STATS=1" python -m lajolla_public_amd.build      (build container; the .so travels)
cd $GRAFT_REPO_ROOT
export LJ_NO_REBUILD=1 LJ_VARIANT=vstats LJ_VOLPATH_STATS=1
for c in vol_cbox_teapot:64 hetvol:64 volpath_test6:64; do
  echo "== ${c%%:*} @ ${c##*:} spp"
  timeout -k 10 300 python3 tools/render_once.py scenes/volpath_test/${c%%:*}.xml ${c##*:} 1 0 2>&1 | grep -v Warning
done
