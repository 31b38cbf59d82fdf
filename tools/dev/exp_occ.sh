cd $GRAFT_REPO_ROOT
for sc in hetvol vol_cbox_teapot volpath_test6; do for o in 2 3 4; do echo -n "$sc occ=$o: "; LJ_TUNE_VOLPATH_OCC=$o timeout -k 10 100 python3 tools/render_once.py scenes/volpath_test/$sc.xml 64 2 0 2>/dev/null | tail -1 | awk '{print $3, $5}'; done; done
export LJ_NO_REBUILD=1
for cfg in "sponza/sponza.xml 256" "disney_bsdf_test/disney_bsdf.xml 256"; do set -- $cfg
  echo -n "$1 default: "; timeout -k 10 200 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1
  for kb in 20 16 14; do echo -n "$1 occ6 lds_kb=$kb: "; LJ_VARIANT=occ6 LJ_TUNE_EXT_LDS_KB=$kb timeout -k 10 200 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1; done
  echo -n "$1 occ6 lds_kb=16 stack 10: "; LJ_VARIANT=occ6 LJ_TUNE_EXT_LDS_KB=16 LJ_TUNE_EXT_STACK=10 timeout -k 10 200 python3 tools/render_once.py scenes/$1 $2 3 0 2>/dev/null | awk '{print $3}' | sort -n | head -1
done
