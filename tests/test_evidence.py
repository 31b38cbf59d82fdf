"""The committed rocprofv3 evidence bench.py quotes (profiles/<round>_counters.json): well-formed, covering every bench workload, and — when
it was measured on the device sources that are checked in — carrying what the roofline fields are computed from.  (bench.py itself refuses
a profile stamped with another hash; this test says so early, as a skip, instead of a bench line without counters.)"""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_counters_cover_the_bench_workloads():
    path = os.path.join(ROOT, "profiles", f"{bench.ROUND}_counters.json")
    assert os.path.exists(path), "run tools/evidence_%s.sh on the GPU box and commit profiles/%s_counters.json" % (bench.ROUND, bench.ROUND)
    t = json.load(open(path))
    assert set(t) >= {"kernel_source_sha", "source", "workloads"}
    want = {"cbox.xml@256": {"k_mega"}, "disney_bsdf.xml@256": {"k_shade"}, "mi.xml@512": {"k_mega"}, "sponza.xml@1024": {"k_extend", "k_shade"}}
    for wl, kernels in want.items():
        assert wl in t["workloads"], wl
        have = {("k_extend" if k.startswith("k_extend") else k) for k in t["workloads"][wl]}
        assert kernels <= have, (wl, have)
        for k, e in t["workloads"][wl].items():
            if k == "k_resolve":
                continue
            assert e["launches"] >= 1 and e["valu_wave_insts"] > 0 and 0 < e["valu_active_lane_frac"] <= 1, (wl, k)
    head = t["workloads"]["cbox.xml@256"]["k_mega"]
    assert head["fetch_bytes"] > 0 and head["write_bytes"] > 0   # the headline kernel's HBM traffic (FETCH_SIZE doubled + WRITE_SIZE)
    if t["kernel_source_sha"] != bench.kernel_source_sha():
        pytest.skip("profiles/%s_counters.json was measured on other device sources: refresh it with tools/evidence_%s.sh" % (bench.ROUND, bench.ROUND))
    prof, why = bench.profiled_counters("sponza.xml@1024")
    assert why is None and "k_extend" in prof and "k_shade" in prof
