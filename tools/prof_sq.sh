#!/bin/bash
# SQ counters of one cbox 64-spp render, one lane / no fused tail so that kernel durations are those of the kernels alone;
# prints per kernel: duration, VALU instructions, VALU issue utilisation = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x t)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LJ_TUNE_LANES=1 LJ_TUNE_TAIL=0
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d gpurun_out/sq_a -- python3 tools/render_once.py scenes/cbox/cbox.xml 64 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/sq_b -- python3 tools/render_once.py scenes/cbox/cbox.xml 64 1 > /dev/null 2>&1
LJ_EXTEND_STATS=1 python3 tools/render_once.py scenes/cbox/cbox.xml 64 1 2>&1 | grep "extend stats" > gpurun_out/sq_summary.txt
python3 - >> gpurun_out/sq_summary.txt <<PY
import pandas as pd, glob
for f in sorted(glob.glob('gpurun_out/sq_[ab]/*/*counter_collection.csv')):
    d=pd.read_csv(f); d['k']=d['Kernel_Name'].str.extract(r'(k_\w+)')
    d['dur']=d['End_Timestamp']-d['Start_Timestamp'] if 'End_Timestamp' in d else 0
    g=d.groupby(['k','Counter_Name'])['Counter_Value'].sum().unstack()
    print(g.T[['k_extend','k_shade']].to_string())
    t=glob.glob(f.replace('counter_collection','kernel_trace'))
    if t:
        k=pd.read_csv(t[0]); k['k']=k['Kernel_Name'].str.extract(r'(k_\w+)'); k['dur']=k['End_Timestamp']-k['Start_Timestamp']
        dur=k.groupby('k')['dur'].sum()
        print((dur/1e6).to_string(), "  (ms, under the counter pass)")
        if 'SQ_INSTS_VALU' in g.columns:
            for kk in ('k_extend','k_shade'):
                print(kk, "VALU issue utilisation %.1f %%" % (100*g.loc[kk,'SQ_INSTS_VALU']*4/(1024*2.4*dur[kk])))
PY
cat gpurun_out/sq_summary.txt
