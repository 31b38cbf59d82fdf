#!/usr/bin/env bash
# TEST INFRASTRUCTURE — builds the Embree-free subset of the reference from its own sources where they
# lie under /root/reference (nothing is copied), into oracle/_ref/ (git-ignored, travels with gpurun).
#
#   oracle/_ref/libljref.so   reference TUs: camera filter image light material shape table_dist transform
#                             parse_obj load_serialized parse_scene scene medium phase_function volume
#                             intersection + 3rdparty/pugixml.cpp + 3rdparty/miniz.c
#   oracle/_ref/gen_golden    oracle/gen_golden.cpp linked against it
#   oracle/_ref/decode_with_reference   oracle/decode_with_reference.cpp: the reference's imread3 / imread1 on tests/assets/images/*
#
# The full reference is UNBUILDABLE here: embree/lib-linux/libembree3.so.3 is listed in .MISSING_LARGE_BLOBS.
# The 20 rtc* symbols stay unresolved in libljref.so (lazy binding); nothing we call reaches them, and we do
# not write stand-ins for them.  g++ only: the reference's RNG draw order at path_tracing.h:11-12 is
# compiler-dependent and "g++ 11" is the build the oracle is defined against.
#
# usage: oracle/ref_build.sh [--golden]    (--golden regenerates tests/golden/*.json)
set -euo pipefail
REF=${LJ_REFERENCE_ROOT:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/src" ]; then echo "ref_build: $REF absent (GPU box?) — nothing to do"; exit 0; fi
mkdir -p "$OUT/obj"
CXXFLAGS="-std=c++17 -O2 -fPIC -w -I$REF/embree/include -I$REF/src"
pids=()
for f in camera filter image light material shape table_dist transform parse_obj load_serialized \
         parse_scene scene medium phase_function volume intersection; do
  if [ ! -f "$OUT/obj/$f.o" ] || [ "$REF/src/$f.cpp" -nt "$OUT/obj/$f.o" ]; then
    g++ $CXXFLAGS -c "$REF/src/$f.cpp" -o "$OUT/obj/$f.o" & pids+=($!)
  fi
done
[ -f "$OUT/obj/pugixml.o" ] || { g++ $CXXFLAGS -c "$REF/src/3rdparty/pugixml.cpp" -o "$OUT/obj/pugixml.o" & pids+=($!); }
[ -f "$OUT/obj/miniz.o" ]   || { gcc -O2 -fPIC -w -c "$REF/src/3rdparty/miniz.c" -o "$OUT/obj/miniz.o" & pids+=($!); }
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
g++ -shared -o "$OUT/libljref.so" "$OUT"/obj/*.o -Wl,-z,lazy -lpthread
g++ $CXXFLAGS -o "$OUT/gen_golden" "$HERE/gen_golden.cpp" -L"$OUT" -lljref \
    -Wl,--allow-shlib-undefined -Wl,-z,lazy -Wl,-rpath,"$OUT" -lpthread
g++ $CXXFLAGS -o "$OUT/decode_with_reference" "$HERE/decode_with_reference.cpp" -L"$OUT" -lljref \
    -Wl,--allow-shlib-undefined -Wl,-z,lazy -Wl,-rpath,"$OUT" -lpthread
echo "ref_build: built $OUT/libljref.so, $OUT/gen_golden and $OUT/decode_with_reference"
if [ "${1:-}" = "--golden" ]; then
  mkdir -p "$HERE/../tests/golden"
  "$OUT/gen_golden" "$REF" "$HERE/../tests/golden"
  "$OUT/decode_with_reference" "$HERE/../tests/golden/image_decode.json" "$HERE"/../tests/assets/images/*
  echo "ref_build: wrote $(ls "$HERE/../tests/golden" | wc -l) fixtures to tests/golden/"
fi
