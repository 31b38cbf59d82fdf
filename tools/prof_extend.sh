#!/bin/bash
# usage (GPU box): tools/prof_extend.sh [scene.xml [spp]] — SQ and TA / TCP counters per kernel of one render (one lane, no fused tail),
# kernel time in ms at the end.  LJ_TUNE_BVH8=0 in the environment profiles the BVH4 extend kernel instead of k_extend8.
SCENE=${1:-scenes/sponza/sponza.xml}; SPP=${2:-32}; TAG=${3:-pmcx}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LJ_TUNE_LANES=1 LJ_TUNE_TAIL=0
rm -rf gpurun_out/${TAG}_a gpurun_out/${TAG}_b gpurun_out/${TAG}_c
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/${TAG}_a -- python3 tools/render_once.py $SCENE $SPP 1 > /dev/null 2>&1
rocprofv3 --pmc TA_TA_BUSY TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES --kernel-trace --output-format csv -d gpurun_out/${TAG}_b -- python3 tools/render_once.py $SCENE $SPP 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/${TAG}_c -- python3 tools/render_once.py $SCENE $SPP 1 > /dev/null 2>&1
python3 - <<PY
import pandas as pd, glob
pd.set_option('display.width', 200)
for f in sorted(glob.glob('gpurun_out/${TAG}_[abc]/*/*counter_collection.csv')):
    d=pd.read_csv(f); d['k']=d['Kernel_Name'].str.extract(r'(k_\w+)')
    g=d.groupby(['k','Counter_Name'])['Counter_Value'].sum().unstack()
    print(g.T.to_string())
t=pd.read_csv(glob.glob('gpurun_out/${TAG}_a/*/*kernel_trace.csv')[0]); t['k']=t['Kernel_Name'].str.extract(r'(k_\w+)')
print((t.groupby('k').apply(lambda x:(x['End_Timestamp']-x['Start_Timestamp']).sum())/1e6).to_string())
PY
