#!/bin/bash
# rocprofv3 kernel stats of bench.py with the render restricted to one lane (LJ_TUNE_LANES=1): launches do not overlap,
# so per-launch durations are those of the kernels alone; LJ_TUNE_TAIL=0 keeps the tail as separate launches too — the
# configuration bench.py's instrumented roofline pass runs in.
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LJ_TUNE_LANES=1 LJ_TUNE_TAIL=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bench1_$TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench1_$TAG.json 2> gpurun_out/bench1_$TAG.err
tail -1 gpurun_out/bench1_$TAG.json | cut -c1-300
cat gpurun_out/bench1_$TAG/*/*kernel_stats.csv | cut -c1-200
