#!/bin/bash
# Everything profiles/r03_* is refreshed from, in one gpurun call:  tools/evidence_r03.sh
#   1. rocprofv3 --kernel-trace --stats of `bench.py` (default workload + configs 3-5)  -> gpurun_out/r03_bench_kernel_stats.csv
#   2. PMC passes, each in its own run with --kernel-trace only, over one render (tools/render_once.py) of every bench workload:
#        SQ set a for all four and two volumetric scenes (hetvol, vol_cbox_teapot @ 64 spp: `bench.py --scene ... --spp 64` quotes them); FETCH_SIZE | WRITE_SIZE | SQ set b for the headline workload   -> gpurun_out/r03_counters.json (stamped with the kernel sources' hash)
#   3. the default bench line (headline + "configs")                                     -> gpurun_out/r03_bench_default_line.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/r03_stats $O/r03_pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/r03_bench_under_rocprof.json 2> $O/r03_bench_under_rocprof.err
cp $O/r03_stats/*/*kernel_stats.csv $O/r03_bench_kernel_stats.csv && cut -c1-150 $O/r03_bench_kernel_stats.csv | head -12 && echo "stats done" || exit 1
SQA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"
SQB="SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU"
i=0
for c in cbox/cbox.xml:256 disney_bsdf_test/disney_bsdf.xml:256 veach_mi/mi.xml:512 sponza/sponza.xml:1024 volpath_test/hetvol.xml:64 volpath_test/vol_cbox_teapot.xml:64; do
  R="python3 tools/render_once.py scenes/${c%%:*} ${c##*:} 1"
  rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $O/r03_pmc_sqa_$i -- $R > /dev/null 2>&1 || exit 1
  echo "${c} sqa done"
  i=$((i+1))
done
R="python3 tools/render_once.py scenes/cbox/cbox.xml 256 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r03_pmc_fetch_0 -- $R > /dev/null 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r03_pmc_write_0 -- $R > /dev/null 2>&1 &&
rocprofv3 --pmc $SQB --kernel-trace --output-format csv -d $O/r03_pmc_sqb_0 -- $R > /dev/null 2>&1 &&
echo "pmc passes done" &&
python3 tools/evidence_collect.py r03 cbox.xml@256 disney_bsdf.xml@256 mi.xml@512 sponza.xml@1024 hetvol.xml@64 vol_cbox_teapot.xml@64 &&
cp $O/r03_counters.json profiles/r03_counters.json &&   # so that the bench line below quotes them (same sources, same box)
timeout -k 10 600 python3 bench.py > $O/r03_bench_default_line.json 2> $O/r03_bench_default.err && echo "default bench done" && cut -c1-400 $O/r03_bench_default_line.json &&
for v in hetvol vol_cbox_teapot; do timeout -k 10 300 python3 bench.py --scene scenes/volpath_test/$v.xml --spp 64 --no-cpu-baseline > $O/r03_bench_$v.json 2>> $O/r03_bench_default.err && cut -c1-200 $O/r03_bench_$v.json; done
