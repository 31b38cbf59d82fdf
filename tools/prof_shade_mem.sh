#!/bin/bash
# usage (GPU box): tools/prof_shade_mem.sh [scene.xml [spp]] — memory-side counters per kernel of one render: HBM bytes (FETCH_SIZE doubled, WRITE_SIZE),
# L2 hits / misses, L1 requests; one rocprofv3 pass per set
SCENE=${1:-scenes/sponza/sponza.xml}; SPP=${2:-64}; TAG=${3:-pmem}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LJ_TUNE_LANES=1 LJ_TUNE_TAIL=0
rm -rf gpurun_out/${TAG}_*
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_a -- python3 tools/render_once.py $SCENE $SPP 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_b -- python3 tools/render_once.py $SCENE $SPP 1 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d gpurun_out/${TAG}_c -- python3 tools/render_once.py $SCENE $SPP 1 > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/${TAG}_d -- python3 tools/render_once.py $SCENE $SPP 1 > /dev/null 2>&1
python3 - <<PY
import pandas as pd, glob
pd.set_option('display.width', 200)
for f in sorted(glob.glob('gpurun_out/${TAG}_[abcd]/*/*counter_collection.csv')):
    d=pd.read_csv(f); d['k']=d['Kernel_Name'].str.extract(r'(k_\w+)')
    g=d.groupby(['k','Counter_Name'])['Counter_Value'].sum().unstack()
    print(g.T.to_string())
t=pd.read_csv(glob.glob('gpurun_out/${TAG}_a/*/*kernel_trace.csv')[0]); t['k']=t['Kernel_Name'].str.extract(r'(k_\w+)'); t['d']=t['End_Timestamp']-t['Start_Timestamp']
print((t.groupby('k')['d'].sum()/1e6).to_string())
PY
rm -rf gpurun_out/${TAG}_*
