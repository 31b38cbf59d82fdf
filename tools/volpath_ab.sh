#!/bin/bash
# A/B of library variants on the volumetric scenes: tools/volpath_ab.sh <variant>...   (best of 2 device ms)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = "default" ]; then unset LJ_VARIANT; else export LJ_VARIANT=$v; fi
  for c in volpath_test2:256 volpath_test4:256 volpath_test6:256 hetvol:16 vol_cbox_teapot:16; do
    echo -n "$v ${c%%:*} spp=${c##*:}: "; timeout -k 10 200 python3 tools/render_once.py scenes/volpath_test/${c%%:*}.xml ${c##*:} 2 0 2>/dev/null | awk '{print $3}' | sort -n | head -1
  done
done
