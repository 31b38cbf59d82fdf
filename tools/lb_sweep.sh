#!/bin/bash
for W in 4 5 6 8; do
  sed -i "s/__launch_bounds__(kBlock, [0-9]) k_extend(/__launch_bounds__(kBlock, $W) k_extend(/" lajolla_public_amd/csrc/device/kernels.hip
  python -m lajolla_public_amd.build 2>&1 | tail -1 > /dev/null
  echo -n "extend waves/SIMD=$W: "; python3 tools/render_once.py scenes/cbox/cbox.xml 256 2 2>/dev/null | tail -1 | cut -c1-60
done
