#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box):  tools/prof_bench.sh <tag>
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m lajolla_public_amd.build --all 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bench_$TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.json | cut -c1-400
cat gpurun_out/bench_$TAG/*/*kernel_stats.csv
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limits), kernel-trace only
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_$TAG -- python3 tools/render_once.py scenes/cbox/cbox.xml 256 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_$TAG -- python3 tools/render_once.py scenes/cbox/cbox.xml 256 1 > /dev/null 2>&1
python3 - <<PY
import pandas as pd, glob, json
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) over tools/render_once.py scenes/cbox/cbox.xml 256 1; counters in KiB; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B for 16 B/lane streaming reads)", "kernels": {}}
for kind in ('fetch','write'):
    f=glob.glob('gpurun_out/pmc_%s_$TAG/*/*counter_collection.csv' % kind)[0]
    d=pd.read_csv(f); d['k']=d['Kernel_Name'].str.extract(r'(k_\w+)')
    g=d.groupby(['k','Counter_Name'])['Counter_Value'].agg(['sum','count'])
    print(g.to_string())
    for (k, c), row in g.iterrows():
        e = out["kernels"].setdefault(k, {})
        e["launches"] = int(row["count"])
        e[kind + "_bytes"] = int(row["sum"] * 1024 * (2 if kind == 'fetch' else 1))
json.dump(out, open('gpurun_out/hbm_traffic_$TAG.json', 'w'), indent=1)
PY
