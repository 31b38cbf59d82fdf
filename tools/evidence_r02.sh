#!/bin/bash
# Everything profiles/r02_* is refreshed from, in one gpurun call:  tools/evidence_r02.sh
#   1. rocprofv3 --kernel-trace --stats of `bench.py` (default workload)            -> gpurun_out/r02_bench_kernel_stats.csv
#   2. PMC passes over the same workload (tools/render_once.py cbox 256 spp), each in its own run with --kernel-trace only:
#        FETCH_SIZE | WRITE_SIZE | SQ set a | SQ set b                               -> gpurun_out/r02_counters.json (stamped with the kernel sources' hash)
#   3. the default bench line and one line per BASELINE config 3 / 4 / 5            -> gpurun_out/r02_bench_*.json(l)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
R="python3 tools/render_once.py scenes/cbox/cbox.xml 256 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/r02_bench_under_rocprof.json 2> $O/r02_bench_under_rocprof.err
cp $O/r02_stats/*/*kernel_stats.csv $O/r02_bench_kernel_stats.csv && cut -c1-150 $O/r02_bench_kernel_stats.csv | head -8 && echo "stats done" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r02_pmc_fetch -- $R > /dev/null 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r02_pmc_write -- $R > /dev/null 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/r02_pmc_sqa -- $R > /dev/null 2>&1 &&
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/r02_pmc_sqb -- $R > /dev/null 2>&1 &&
echo "pmc passes done" &&
python3 - <<'PY'
import glob, hashlib, json, os
import pandas as pd
root = os.environ.get("GRAFT_REPO_ROOT", ".")
h = hashlib.sha256()
d = os.path.join(root, "lajolla_public_amd", "csrc", "device")
for f in sorted(os.listdir(d)):
    h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
out = {"kernel_source_sha": h.hexdigest()[:16],
       "source": "rocprofv3 --pmc <set> --kernel-trace, one run per set (FETCH_SIZE | WRITE_SIZE | two SQ sets), over tools/render_once.py scenes/cbox/cbox.xml 256 1 "
                 "(the bench workload, one render); FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE doubled (gfx950 tallies 128-B requests of 16 B/lane streams as 64 B, "
                 "MI355X_MICROARCH.md section HBM); totals over the render's launches of each kernel",
       "kernels": {}}
for tag in ("fetch", "write", "sqa", "sqb"):
    f = glob.glob(f"gpurun_out/r02_pmc_{tag}/*/*counter_collection.csv")[0]
    df = pd.read_csv(f); df["k"] = df["Kernel_Name"].str.extract(r"(k_\w+)")
    # one row per (dispatch, counter): launches = dispatches of the kernel in this pass
    for (k, c), g in df.groupby(["k", "Counter_Name"]):
        e = out["kernels"].setdefault(k, {})
        e["launches"] = int(g["Dispatch_Id"].nunique()) if "Dispatch_Id" in g else int(len(g))
        v = float(g["Counter_Value"].sum())
        if c == "FETCH_SIZE": e["fetch_bytes"] = int(v * 1024 * 2)
        elif c == "WRITE_SIZE": e["write_bytes"] = int(v * 1024)
        else: e[c] = v
    t = glob.glob(f.replace("counter_collection", "kernel_trace"))
    if t and tag == "sqa":
        kt = pd.read_csv(t[0]); kt["k"] = kt["Kernel_Name"].str.extract(r"(k_\w+)"); kt["dur"] = kt["End_Timestamp"] - kt["Start_Timestamp"]
        for k, g in kt.groupby("k"):
            out["kernels"].setdefault(k, {})["ns_under_sq_pass"] = int(g["dur"].sum())
PEAK_WAVE_INST_PER_S = 1024 * 2.4e9 / 2   # 256 CUs x 4 SIMD-32, one wave64 VALU instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)
for k, e in out["kernels"].items():
    if "SQ_INSTS_VALU" in e and e.get("ns_under_sq_pass"):
        e["valu_wave_insts"] = e["SQ_INSTS_VALU"]
        e["valu_issue_frac"] = round(e["SQ_INSTS_VALU"] / (e["ns_under_sq_pass"] * 1e-9) / PEAK_WAVE_INST_PER_S, 4)
        if e.get("SQ_THREAD_CYCLES_VALU"): e["valu_active_lane_frac"] = round(e["SQ_THREAD_CYCLES_VALU"] / e["SQ_INSTS_VALU"] / 64.0, 4)
json.dump(out, open("gpurun_out/r02_counters.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: {x: e[x] for x in ("launches", "fetch_bytes", "write_bytes", "valu_issue_frac", "valu_active_lane_frac") if x in e} for k, e in out["kernels"].items()}, indent=1))
PY
cp $O/r02_counters.json profiles/r02_counters.json 2>/dev/null   # so that the bench lines below quote them (same sources, same box)
timeout -k 10 400 python3 bench.py > $O/r02_bench_default_line.json 2> $O/r02_bench_default.err && echo "default bench done" && cut -c1-260 $O/r02_bench_default_line.json &&
: > $O/r02_bench_configs_3_4_5.jsonl &&
for c in disney_bsdf_test/disney_bsdf.xml:256 veach_mi/mi.xml:512 sponza/sponza.xml:1024; do
  timeout -k 10 300 python3 bench.py --scene scenes/${c%%:*} --spp ${c##*:} --steps 3 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/r02_bench_configs_3_4_5.jsonl && echo "config ${c%%:*} done" || exit 1
done
cut -c1-200 $O/r02_bench_configs_3_4_5.jsonl
cp profiles/r02_counters.json $O/r02_counters_copy.json 2>/dev/null
