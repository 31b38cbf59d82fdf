#!/bin/bash
# the child ordering of a BVH4 node step, variant libraries against the default one (built beforehand: LJ_VARIANT=mm|s4|s3, see
# profiles/r03_sweeps.txt): best-of-3 device ms per render, two interleaved rounds; the extend kernel's step counters of each at the end
cd $GRAFT_REPO_ROOT
export LJ_NO_REBUILD=1
run() { timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; }
VARIANTS=${VARIANTS:-"default mm s4 s3"}
for cfg in "sponza/sponza.xml 256" "volpath_test/vol_cbox_teapot.xml 64"; do set -- $cfg
  [ -f scenes/$1 ] || continue
  for round in 1 2; do for v in $VARIANTS; do
    if [ "$v" = "default" ]; then unset LJ_VARIANT; else export LJ_VARIANT=$v; fi
    echo -n "$1 @ $2 $v: "; run $1 $2
  done; done
done
for v in $VARIANTS; do
  if [ "$v" = "default" ]; then unset LJ_VARIANT; else export LJ_VARIANT=$v; fi
  echo -n "$v "; LJ_EXTEND_STATS=1 timeout -k 10 200 python3 tools/render_once.py scenes/sponza/sponza.xml 64 1 0 2>&1 | grep "extend stats"
done
