#!/bin/bash
# best-of-3 device ms per render for the four bench scenes (quick regression check of a kernel change)
cd $GRAFT_REPO_ROOT
for s in cbox/cbox.xml:256 sponza/sponza.xml:64 disney_bsdf_test/disney_bsdf.xml:64 veach_mi/mi.xml:256; do echo -n "${s%%:*}: "; timeout -k 10 200 python3 tools/render_once.py scenes/${s%%:*} ${s##*:} 3 2>/dev/null | awk '{print $3}' | sort -n | head -1; done
