#!/bin/bash
# kernel trace of two default cbox renders (timeline analysis: tools/trace_timeline.py gpurun_out/trace_render)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_render
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_render -- python3 tools/render_once.py ${1:-scenes/cbox/cbox.xml} ${2:-256} 3 > gpurun_out/trace_render.log 2>&1
tail -2 gpurun_out/trace_render.log
