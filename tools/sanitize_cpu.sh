#!/bin/bash
# AddressSanitizer + UBSan run of the CPU-side test libraries (oracle/lj_oracle.cpp and the host twin of the device headers with the
# product's flatten.cpp / bvh.cpp) over the part of the CPU suite that drives them.  Build container or GPU box; no GPU needed.
#   tools/sanitize_cpu.sh [pytest args]        -> profiles/r02_sanitizer.txt holds the last run
cd "$(dirname "$0")/.."
export LJ_SANITIZE=1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
python3 -m pytest tests/test_oracle_golden.py tests/test_twin_parity.py tests/test_device_kats.py tests/test_abi_and_host.py tests/test_aux_integrators.py tests/test_volpath.py \
    -q -m "not gpu" -p no:cacheprovider "$@" 2>&1 | tail -15
