#!/bin/bash
# usage (GPU box): tools/prof_sponza.sh [scene.xml [spp]] — is the extend kernel on sponza (or the given scene) bound by VALU issue or by the vector L1 / TA gather path?
SCENE=${1:-scenes/sponza/sponza.xml}; SPP=${2:-32}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LJ_TUNE_LANES=1 LJ_TUNE_TAIL=0
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmcz_a -- python3 tools/render_once.py $SCENE $SPP 1 > /dev/null 2>&1
rocprofv3 --pmc TA_TA_BUSY TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcz_b -- python3 tools/render_once.py $SCENE $SPP 1 > /dev/null 2>&1
python3 - <<PY
import pandas as pd, glob
for f in sorted(glob.glob('gpurun_out/pmcz_*/*/*counter_collection.csv')):
    d=pd.read_csv(f); d['k']=d['Kernel_Name'].str.extract(r'(k_\w+)')
    g=d.groupby(['k','Counter_Name'])['Counter_Value'].sum().unstack()
    print(g.T[['k_extend','k_shade']].to_string())
t=pd.read_csv(glob.glob('gpurun_out/pmcz_a/*/*kernel_trace.csv')[0]); t['k']=t['Kernel_Name'].str.extract(r'(k_\w+)')
print((t.groupby('k').apply(lambda x:(x['End_Timestamp']-x['Start_Timestamp']).sum())/1e6).to_string())
PY
