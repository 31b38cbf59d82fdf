#!/bin/bash
# shade-side knobs on the bench workloads: image descriptors + environment-map marginal tables staged in LDS (LJ_TUNE_STAGE_IMAGES=0: left in global memory)
cd $GRAFT_REPO_ROOT
for cfg in "disney_bsdf_test/disney_bsdf.xml 256" "sponza/sponza.xml 256" "veach_mi/mi.xml 512" "cbox/cbox.xml 256"; do set -- $cfg
  for v in 1 0; do
    echo "== $1 spp=$2 stage_images=$v"
    LJ_TUNE_STAGE_IMAGES=$v timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | tail -2
    LJ_TUNE_STAGE_IMAGES=$v timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 1 1 2>/dev/null | tail -1
  done
done
