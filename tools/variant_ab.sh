#!/bin/bash
# A/B of library variants on the large scenes: tools/variant_ab.sh <variant> ...   ("default" = the shipped build); best of 3 device ms
cd $GRAFT_REPO_ROOT
export LJ_NO_REBUILD=1
run() { timeout -k 10 300 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; }
for cfg in "sponza/sponza.xml 64" "sponza/sponza.xml 256" "disney_bsdf_test/disney_bsdf.xml 64" "disney_bsdf_test/disney_bsdf.xml 256"; do set -- $cfg
  for v in "$@"; do :; done
  echo "== $1 spp=$2"
  for v in $VARIANTS; do
    if [ "$v" = "default" ]; then unset LJ_VARIANT; else export LJ_VARIANT=$v; fi
    echo -n "$v: "; run $1 $2
  done
done
