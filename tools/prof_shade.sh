#!/bin/bash
# usage (GPU box): tools/prof_shade.sh  — SQ counters of one cbox 64-spp render
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmcs_a -- python3 tools/render_once.py scenes/cbox/cbox.xml 64 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/pmcs_b -- python3 tools/render_once.py scenes/cbox/cbox.xml 64 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d gpurun_out/pmcs_c -- python3 tools/render_once.py scenes/cbox/cbox.xml 64 1 > /dev/null 2>&1
python3 - <<PY
import pandas as pd, glob
for f in sorted(glob.glob('gpurun_out/pmcs_*/*/*counter_collection.csv')):
    d=pd.read_csv(f); d['k']=d['Kernel_Name'].str.extract(r'(k_\w+)')
    g=d.groupby(['k','Counter_Name'])['Counter_Value'].sum().unstack()
    print(g.T[['k_extend','k_shade']].to_string())
PY
