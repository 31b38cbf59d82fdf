#!/bin/bash
# bench lines + rocprofv3 kernel stats of the volumetric path tracer on the reference's volpath scenes (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
: > $O/r02_bench_volpath.jsonl
for c in volpath_test1:256 volpath_test2:256 volpath_test4:256 volpath_test5:256 volpath_test6:256 hetvol:64 vol_cbox_teapot:64; do
  timeout -k 10 300 python3 bench.py --scene scenes/volpath_test/${c%%:*}.xml --spp ${c##*:} --steps 3 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/r02_bench_volpath.jsonl || echo "FAILED $c"
done
python3 -c "
import json
for l in open('$O/r02_bench_volpath.jsonl'):
    d = json.loads(l); print(d['metric'], d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_volpath_stats -- python3 tools/render_once.py scenes/volpath_test/volpath_test6.xml 256 2 > $O/r02_volpath_stats.log 2>&1
cp $O/r02_volpath_stats/*/*kernel_stats.csv $O/r02_volpath_kernel_stats.csv; cut -c1-160 $O/r02_volpath_kernel_stats.csv | head -5
