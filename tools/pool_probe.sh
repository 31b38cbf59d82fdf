#!/bin/bash
# pooled leaf phase of k_extend: parity first, then large-scene timing over the hand-over threshold (LJ_TUNE_MINDESC)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/pool_tests.log 2>&1 || { tail -30 gpurun_out/pool_tests.log; exit 1; }
tail -3 gpurun_out/pool_tests.log
for s in "disney_bsdf_test/disney_bsdf.xml 64" "sponza/sponza.xml 64"; do set -- $s
  for md in 16 24 32 40 48; do echo -n "$1 spp=$2 mindesc=$md: "; LJ_TUNE_MINDESC=$md timeout -k 10 200 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; done
  LJ_EXTEND_STATS=1 timeout -k 10 200 python3 tools/render_once.py scenes/$1 $2 1 0 2>&1 | grep "extend stats"
done
