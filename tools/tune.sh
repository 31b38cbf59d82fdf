#!/bin/bash
python -m lajolla_public_amd.build 2>&1 | tail -1
for r in 8 16 24 32 48 64; do for m in 1 8 16 24; do
  echo -n "refill=$r mindesc=$m: "; LJ_TUNE_REFILL=$r LJ_TUNE_MINDESC=$m python3 tools/render_once.py scenes/cbox/cbox.xml 256 2 2>/dev/null | tail -1 | cut -c1-60
done; done
