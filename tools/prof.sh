#!/bin/bash
# usage: tools/prof.sh <tag> [scene] [spp]   (run on the GPU box via gpurun)
TAG=$1; SCENE=${2:-scenes/cbox/cbox.xml}; SPP=${3:-256}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m lajolla_public_amd.build 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 tools/render_once.py $SCENE $SPP 2 > gpurun_out/prof_$TAG.log 2>&1
grep spp gpurun_out/prof_$TAG.log
cat gpurun_out/prof_$TAG/*/*kernel_stats.csv | cut -c1-150
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_a -- python3 tools/render_once.py $SCENE 64 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_b -- python3 tools/render_once.py $SCENE 64 1 > /dev/null 2>&1
python3 - <<PY
import pandas as pd, glob
for f in sorted(glob.glob('gpurun_out/pmc_${TAG}_*/*/*counter_collection.csv')):
    d=pd.read_csv(f); d['k']=d['Kernel_Name'].str.extract(r'(k_\w+)')
    g=d.groupby(['k','Counter_Name'])['Counter_Value'].sum().unstack()
    print(g.T[['k_extend','k_shade']].to_string())
PY
