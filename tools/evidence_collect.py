"""Folds the rocprofv3 PMC passes of tools/evidence_rNN.sh into gpurun_out/<round>_counters.json: per bench workload and kernel the
counter totals over one render's launches, launches, duration under the SQ pass, VALU issue fraction and active-lane fraction;
stamped with the hash of the device sources so that bench.py never quotes a profile of other kernels.
    python3 tools/evidence_collect.py r03 cbox.xml@256 disney_bsdf.xml@256 ...   (workload i <- gpurun_out/<round>_pmc_<set>_<i>)"""
import glob, hashlib, json, os, sys
import pandas as pd

rnd, workloads = sys.argv[1], sys.argv[2:]
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
h = hashlib.sha256()
d = os.path.join(root, "lajolla_public_amd", "csrc", "device")
for f in sorted(os.listdir(d)):
    if not os.path.isfile(os.path.join(d, f)) or f.startswith("."):
        continue
    h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
out = {"kernel_source_sha": h.hexdigest()[:16],
       "source": "rocprofv3 --pmc <set> --kernel-trace, one run per set and workload (SQ set a for every bench workload; FETCH_SIZE | WRITE_SIZE | SQ set b for the headline "
                 "workload), each over tools/render_once.py <scene> <spp> 1 (one render); FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE doubled (gfx950 tallies 128-B requests of "
                 "16 B/lane streams as 64 B, MI355X_MICROARCH.md section HBM); totals over the render's launches of each kernel",
       "workloads": {}}
PEAK_WAVE_INST_PER_S = 1024 * 2.4e9 / 2   # 256 CUs x 4 SIMD-32, one wave64 VALU instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)
for i, wl in enumerate(workloads):
    kernels = out["workloads"].setdefault(wl, {})
    for tag in ("fetch", "write", "sqa", "sqb"):
        fs = glob.glob(os.path.join(root, f"gpurun_out/{rnd}_pmc_{tag}_{i}/*/*counter_collection.csv"))
        if not fs:
            continue
        df = pd.read_csv(fs[0]); df["k"] = df["Kernel_Name"].str.extract(r"(k_\w+)")
        for (k, c), g in df.groupby(["k", "Counter_Name"]):
            e = kernels.setdefault(k, {})
            e["launches"] = int(g["Dispatch_Id"].nunique()) if "Dispatch_Id" in g else int(len(g))
            v = float(g["Counter_Value"].sum())
            if c == "FETCH_SIZE": e["fetch_bytes"] = int(v * 1024 * 2)
            elif c == "WRITE_SIZE": e["write_bytes"] = int(v * 1024)
            else: e[c] = v
        t = glob.glob(fs[0].replace("counter_collection", "kernel_trace"))
        if t and tag == "sqa":
            kt = pd.read_csv(t[0]); kt["k"] = kt["Kernel_Name"].str.extract(r"(k_\w+)"); kt["dur"] = kt["End_Timestamp"] - kt["Start_Timestamp"]
            for k, g in kt.groupby("k"):
                kernels.setdefault(k, {})["ns_under_sq_pass"] = int(g["dur"].sum())
    for k, e in kernels.items():
        if "SQ_INSTS_VALU" in e and e.get("ns_under_sq_pass"):
            e["valu_wave_insts"] = e["SQ_INSTS_VALU"]
            e["valu_issue_frac"] = round(e["SQ_INSTS_VALU"] / (e["ns_under_sq_pass"] * 1e-9) / PEAK_WAVE_INST_PER_S, 4)
            if e.get("SQ_THREAD_CYCLES_VALU"): e["valu_active_lane_frac"] = round(e["SQ_THREAD_CYCLES_VALU"] / e["SQ_INSTS_VALU"] / 64.0, 4)
            if e.get("SQ_WAVE_CYCLES") and e.get("SQ_WAIT_ANY"): e["wave_time_waiting_frac"] = round(e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"], 4)
            # the issue floor bench.py prices `floor_frac` against: the instructions the launch would issue if every one of
            # them ran with all 64 lanes live — SQ_THREAD_CYCLES_VALU / 64 — at the chip's issue peak; a hand-derived count overrides it below
            if e.get("SQ_THREAD_CYCLES_VALU"): e["floor_wave_insts"] = e["SQ_THREAD_CYCLES_VALU"] / 64.0; e["floor_kind"] = "lane-weighted executed instructions"
# instruction floors derived by hand ride along when present
floors = os.path.join(root, "profiles", f"{rnd}_floors.json")
if os.path.exists(floors):
    for wl, ks in json.load(open(floors)).items():
        for k, v in ks.items():
            if wl in out["workloads"] and k in out["workloads"][wl]:
                out["workloads"][wl][k]["floor_wave_insts"] = v; out["workloads"][wl][k]["floor_kind"] = "hand-derived (DESIGN.md section 4)"
json.dump(out, open(os.path.join(root, f"gpurun_out/{rnd}_counters.json"), "w"), indent=1, sort_keys=True)
keep = ("launches", "fetch_bytes", "write_bytes", "valu_issue_frac", "valu_active_lane_frac", "wave_time_waiting_frac")
print(json.dumps({wl: {k: {x: e[x] for x in keep if x in e} for k, e in ks.items()} for wl, ks in out["workloads"].items()}, indent=1))
