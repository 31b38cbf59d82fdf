#!/bin/bash
# GPU-box check with a record: runs the given pytest selection (default: the whole `-m gpu` suite) with -rA and keeps the full report under
# gpurun_out/, so that a red run names its test.   tools/gpu_check.sh [name] [pytest args...]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
name=${1:-gputests}; shift
if [ $# -eq 0 ]; then set -- tests -m gpu; fi
timeout -k 10 900 python3 -m pytest "$@" -q -rA -p no:cacheprovider > gpurun_out/$name.log 2>&1
rc=$?
grep -E "^(FAILED|ERROR)|passed|failed|error" gpurun_out/$name.log | tail -20
exit $rc
