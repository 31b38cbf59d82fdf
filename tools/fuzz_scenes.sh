#!/bin/bash
# scene front end under ASan + UBSan on damaged copies of small shipped scenes:  tools/fuzz_scenes.sh [rounds per file, default 150]
cd "$(dirname "$0")/.."
H=lajolla_public_amd/csrc/host
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -o /tmp/lj_fuzz_scenes tools/fuzz/fuzz_scenes.cpp \
    $H/scene_xml.cpp $H/mesh_io.cpp $H/api_host.cpp $H/flatten.cpp $H/bvh.cpp $H/image_io.cpp $H/jpeg_decode.cpp $H/png_decode.cpp $H/tga_bmp_decode.cpp $H/exr_decode.cpp -lz || exit 1
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:allocator_may_return_null=1:max_allocation_size_mb=8192 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
W=/tmp/lj_fuzz_scenes_work; rm -rf $W; mkdir -p $W; cp -r scenes/cbox scenes/veach_mi scenes/disney_bsdf_test scenes/matpreview $W/; mkdir -p $W/volpath_test; cp scenes/volpath_test/{volpath_test1.xml,volpath_test5.xml,hetvol.xml,smoke.vol,bounds.obj,plane.obj} $W/volpath_test/ 2>/dev/null; cp -r scenes/volpath_test/meshes $W/volpath_test/ 2>/dev/null
R=${1:-150}
timeout 3000 /tmp/lj_fuzz_scenes $R $W/cbox cbox.xml cbox.xml $(cd $W/cbox && ls meshes/*.obj | head -2) 2>&1 | tail -12
timeout 3000 /tmp/lj_fuzz_scenes $R $W/veach_mi mi.xml mi.xml 2>&1 | tail -12
timeout 3000 /tmp/lj_fuzz_scenes $R $W/disney_bsdf_test simple_sphere.xml simple_sphere.xml 2>&1 | tail -12
timeout 3000 /tmp/lj_fuzz_scenes $R $W/volpath_test volpath_test5.xml volpath_test5.xml 2>&1 | tail -12
timeout 3000 /tmp/lj_fuzz_scenes $((R / 5 + 1)) $W/volpath_test hetvol.xml hetvol.xml smoke.vol 2>&1 | tail -12
timeout 3000 /tmp/lj_fuzz_scenes $((R / 10 + 1)) $W/matpreview matpreview.xml matpreview.xml matpreview.serialized envmap.exr 2>&1 | tail -12
