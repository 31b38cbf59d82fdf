#!/bin/bash
# SQ / LDS / memory counters of the fused kernel on one cbox render (256 spp), three counter passes + a kernel trace with --stats:
#   tools/prof_mega.sh <tag> [variant]
TAG=$1; [ -n "$2" ] && export LJ_VARIANT=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R="python3 tools/render_once.py scenes/cbox/cbox.xml 256 2"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mega_stats_$TAG -- $R > gpurun_out/mega_stats_$TAG.log 2>&1
cat gpurun_out/mega_stats_$TAG/*/*kernel_stats.csv | cut -c1-160
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/mega_sq_a_$TAG -- $R > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/mega_sq_b_$TAG -- $R > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d gpurun_out/mega_sq_c_$TAG -- $R > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/mega_fetch_$TAG -- $R > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/mega_write_$TAG -- $R > /dev/null 2>&1
python3 - <<PY | tee gpurun_out/mega_counters_$TAG.txt
import pandas as pd, glob
for f in sorted(glob.glob('gpurun_out/mega_*_$TAG/*/*counter_collection.csv')):
    d = pd.read_csv(f); d['k'] = d['Kernel_Name'].str.extract(r'(k_\w+)')
    g = d.groupby(['k', 'Counter_Name'])['Counter_Value'].agg(['sum', 'count'])
    print(g.to_string())
    t = glob.glob(f.replace('counter_collection', 'kernel_trace'))
    if t:
        k = pd.read_csv(t[0]); k['k'] = k['Kernel_Name'].str.extract(r'(k_\w+)'); k['dur'] = k['End_Timestamp'] - k['Start_Timestamp']
        print((k.groupby('k')['dur'].agg(['sum', 'count'])).to_string(), " (ns under the counter pass)")
PY
