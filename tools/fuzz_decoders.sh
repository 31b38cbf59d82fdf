#!/bin/bash
# texture decoders under ASan + UBSan on tests/assets/images/* and damaged copies of them:  tools/fuzz_decoders.sh [rounds per file, default 300]
cd "$(dirname "$0")/.."
H=lajolla_public_amd/csrc/host
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -D_GLIBCXX_SANITIZE_VECTOR -fno-omit-frame-pointer -o /tmp/lj_fuzz_decoders tools/fuzz/fuzz_decoders.cpp \
    $H/image_io.cpp $H/jpeg_decode.cpp $H/png_decode.cpp $H/tga_bmp_decode.cpp $H/exr_decode.cpp -lz || exit 1
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:allocator_may_return_null=1:max_allocation_size_mb=4096 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    timeout 3000 /tmp/lj_fuzz_decoders ${1:-300} tests/assets/images/* $(find scenes -name '*.jpg' -size -300k | head -3) $(find scenes -name '*.exr' -size -1500k | head -2) 2>&1 | tail -25
