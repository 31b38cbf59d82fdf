#!/bin/bash
# sweep of the extend kernel's run-time knobs (refill threshold x leaf hand-over threshold), cbox 256 spp, one lane so that
# the extend time is that of the kernel alone
cd $GRAFT_REPO_ROOT
for r in 4 8 12 16 24 32; do for m in 1 4 16; do
  echo -n "refill=$r mindesc=$m: "; LJ_TUNE_LANES=1 LJ_TUNE_TAIL=0 LJ_TUNE_REFILL=$r LJ_TUNE_MINDESC=$m timeout -k 10 120 python3 tools/render_once.py scenes/cbox/cbox.xml 256 2 1 2>/dev/null | tail -1 | sed 's/.*extend/extend/'
done; done
