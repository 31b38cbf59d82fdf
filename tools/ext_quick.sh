#!/bin/bash
# extend kernel, large scenes: best-of-3 device ms at 64 spp per hand-over threshold + the utilisation counters (developer probe)
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python3 tools/render_once.py scenes/$1 64 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; }
for sc in sponza/sponza.xml disney_bsdf_test/disney_bsdf.xml; do
  echo "== $sc"
  for md in ${MDS:-24 32 40}; do echo -n "min_descending=$md: "; LJ_TUNE_MINDESC=$md run $sc; done
  LJ_EXTEND_STATS=1 timeout -k 10 200 python3 tools/render_once.py scenes/$sc 64 1 0 2>&1 | grep "extend stats"
  timeout -k 10 200 python3 tools/render_once.py scenes/$sc 64 2 1 2>/dev/null | tail -1
done
