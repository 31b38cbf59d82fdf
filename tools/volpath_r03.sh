#!/bin/bash
# the volumetric path tracer (k_volpath: persistent waves, path regeneration) on the reference's volpath scenes: device ms / Msamples/s, best of 3
cd $GRAFT_REPO_ROOT
for c in volpath_test1:256 volpath_test2:256 volpath_test4:256 volpath_test5:256 volpath_test6:256 hetvol:64 vol_cbox_teapot:64 hetvol_colored:64; do
  echo -n "${c%%:*} @ ${c##*:} spp: "
  timeout -k 10 300 python3 tools/render_once.py scenes/volpath_test/${c%%:*}.xml ${c##*:} 3 0 2>/dev/null | awk '{print $3, "ms", $5, "Msamples/s"}' | sort -n | head -1
done
