#!/bin/bash
# The extend kernel's gather-path counters on sponza (32 spp, one render), BVH4 kernel (the default for this scene) against the BVH8 kernel
# (LJ_TUNE_BVH8=1): vector-memory read instructions, TCP->TCC read requests, TA busy, VALU, per ray.   -> gpurun_out/r03_sponza_counters.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SCENE=scenes/sponza/sponza.xml; SPP=32; O=gpurun_out
export LJ_TUNE_LANES=1 LJ_TUNE_TAIL=0
for w in 0 1; do
  export LJ_TUNE_BVH8=$w
  rm -rf $O/spz_a_$w $O/spz_b_$w
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/spz_a_$w -- python3 tools/render_once.py $SCENE $SPP 1 > /dev/null 2>&1 || exit 1
  rocprofv3 --pmc TA_TA_BUSY TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES --kernel-trace --output-format csv -d $O/spz_b_$w -- python3 tools/render_once.py $SCENE $SPP 1 > /dev/null 2>&1 || exit 1
  LJ_EXTEND_STATS=1 timeout -k 10 200 python3 tools/render_once.py $SCENE $SPP 1 0 2>&1 | grep "extend stats" > $O/spz_stats_$w.txt
done
python3 - > $O/r03_sponza_counters.txt <<'PY'
import pandas as pd, glob, re
print("sponza 768x575 @ 32 spp, one render (tools/sponza_counters_r03.sh): the extend kernel's counters, totals over its launches and per ray")
for w, name in ((0, "BVH4 kernel k_extend (default for this scene)"), (1, "BVH8 kernel k_extend8 (LJ_TUNE_BVH8=1)")):
    st = open(f"gpurun_out/spz_stats_{w}.txt").read().strip()
    rays = float(re.search(r"rays (\d+)", st).group(1))
    print(f"\n== {name}\n{st}")
    tot = {}
    for tag in "ab":
        f = glob.glob(f"gpurun_out/spz_{tag}_{w}/*/*counter_collection.csv")[0]
        d = pd.read_csv(f); d["k"] = d["Kernel_Name"].str.extract(r"(k_\w+)")[0].fillna("")
        g = d[d["k"].str.startswith("k_extend")].groupby("Counter_Name")["Counter_Value"].sum()
        for c, v in g.items(): tot[c] = v
    t = pd.read_csv(glob.glob(f"gpurun_out/spz_a_{w}/*/*kernel_trace.csv")[0]); t["k"] = t["Kernel_Name"].str.extract(r"(k_\w+)")[0].fillna("")
    ms = (t[t["k"].str.startswith("k_extend")].eval("End_Timestamp - Start_Timestamp").sum()) / 1e6
    print(f"extend launches, under the SQ pass: {ms:.2f} ms; rays {rays:.0f}")
    for c in sorted(tot): print(f"  {c:34s} {tot[c]:16.0f}   per ray {tot[c] / rays:10.3f}")
    if "SQ_INSTS_VALU" in tot:
        print(f"  VALU issue {tot['SQ_INSTS_VALU'] / (ms * 1e-3) / 1228.8e9:.3f} of peak, lanes active {tot['SQ_THREAD_CYCLES_VALU'] / tot['SQ_INSTS_VALU'] / 64:.3f}, waves waiting {tot['SQ_WAIT_ANY'] / tot['SQ_WAVE_CYCLES']:.3f}")
PY
cat $O/r03_sponza_counters.txt
