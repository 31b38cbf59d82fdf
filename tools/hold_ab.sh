#!/bin/bash
# extend kernel: held leaves on / off / not for shadow rays (library variants built beforehand: nohold, noshadow) x the hand-over threshold
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python3 tools/render_once.py scenes/$1 64 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; }
for sc in sponza/sponza.xml disney_bsdf_test/disney_bsdf.xml; do
  echo "== $sc"
  for v in default noshadow nohold; do
    if [ "$v" = "default" ]; then unset LJ_VARIANT; else export LJ_VARIANT=$v; fi
    export LJ_NO_REBUILD=1
    for md in 16 24 32 40 48; do echo -n "$v min_descending=$md: "; LJ_TUNE_MINDESC=$md run $sc; done
    LJ_EXTEND_STATS=1 timeout -k 10 200 python3 tools/render_once.py scenes/$sc 64 1 0 2>&1 | grep "extend stats"
  done
done
