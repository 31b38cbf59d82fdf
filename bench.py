#!/usr/bin/env python3
"""bench.py — Msamples/s of the path-tracing hot path on MI355X.

One "step" = one complete render of BASELINE.json config 2: scenes/cbox/cbox.xml at 512x512, 256 spp
(67,108,864 camera samples; Lambertian + area light; max_depth -1, Russian roulette from depth 5), scene already
resident in HBM, timed from the first kernel to the resolved framebuffer (and, for N > 1, the RCCL reduce onto
rank 0) — the region the reference times at main.cpp:40-42.  N > 1: one process per GPU (torch.distributed.run), the
image's 16x16 tiles are dealt round-robin to the ranks, each rank renders its tiles at full spp, one sum-reduce.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6.3 TB/s is what a copy kernel achieves


def host_cores():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def profiled_traffic(kernel, launches):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same workload
    (profiles/r01_hbm_traffic.json, written by tools/prof_bench.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md).
    The counters are totals over one render — the same bytes however the render is cut into launches — so they are
    divided by the launch count of the instrumented pass the roofline figures come from."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
        k = t["kernels"][kernel]
        return int((k["fetch_bytes"] + k["write_bytes"]) / max(launches, 1)), t["source"]
    except (OSError, KeyError, ValueError):
        return None, None


def cpu_baseline(scene_xml, seconds_budget=20.0):
    """The CPU restatement of the reference's parallel.cpp tile path (oracle/, `port`), all host cores, on a bounded
    sample of the same workload: cbox 512x512 at a reduced spp chosen to take roughly `seconds_budget`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lajolla_public_amd as lj
    from helpers import Oracle
    hs = lj.parse_scene(scene_xml)
    o = Oracle(hs)
    cores = host_cores()
    rc, _, _, st = o.render(spp=1, rng_mode=1, threads=cores)  # probe
    rate = st.samples / max(st.seconds, 1e-9)
    spp = int(max(1, min(256, round(seconds_budget * rate / (hs.width * hs.height)))))
    rc, _, _, st = o.render(spp=spp, rng_mode=1, threads=cores)
    return {"value": round(st.samples / st.seconds / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"cbox 512x512 at {spp} spp ({st.samples} samples, {st.seconds:.1f} s), reference RNG schedule (one pcg32 stream per 16x16 tile), "
                      f"double precision, {cores} threads pulling tiles from a shared counter"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "cbox", "cbox.xml"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pool", type=int, default=0)
    args = ap.parse_args()

    import torch
    import lajolla_public_amd as lj
    from lajolla_public_amd import dist as ljdist

    rank, world, local_rank = ljdist.env_rank_world()
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible — the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        ljdist.init_process_group("nccl")  # RCCL on ROCm

    hs = lj.parse_scene(args.scene)
    ctx = lj.Context(local_rank)
    scene = lj.Scene(ctx, hs)
    w, h, spp = hs.width, hs.height, args.spp
    total_samples = w * h * spp
    frame = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        # renders this rank's tiles into `frame` on torch's current stream, then the framebuffer reduce
        lj.render_device(scene, frame.data_ptr(), stream=stream.cuda_stream, spp=spp, rank=rank, world_size=world, pool_paths=args.pool)
        ljdist.reduce_framebuffer(frame, dst=0)

    for _ in range(args.warmup):
        step()
    ljdist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ljdist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    elapsed = ljdist.max_over_ranks(elapsed, device=dev)
    ms_per_step = elapsed / args.steps * 1e3
    value = total_samples * args.steps / elapsed / 1e6
    st = scene.stats()
    k_mean = st.bounce_iterations / max(st.samples, 1)

    # ---- roofline of the dominant kernel: one more, instrumented, render (HIP events on the render stream around every
    # extend / shade launch — flags=1).  Kept outside the timed region because the per-step event synchronisation
    # perturbs it; the kernels and their inputs are identical.
    lj.render_device(scene, frame.data_ptr(), stream=stream.cuda_stream, spp=spp, rank=rank, world_size=world, pool_paths=args.pool, flags=1)
    torch.cuda.synchronize(dev)
    si = scene.stats()
    kernels = {
        "k_extend": {"ms": si.extend_ms, "launches": si.extend_launches, "bytes": si.extend_bytes},
        "k_shade": {"ms": si.shade_ms, "launches": si.shade_launches, "bytes": si.shade_bytes},
    }
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    kd = kernels[dom]
    avg_us = kd["ms"] * 1e3 / max(kd["launches"], 1)
    achieved = (kd["bytes"] / max(kd["launches"], 1)) / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
    is_default = os.path.basename(args.scene) == "cbox.xml" and spp == 256 and world == 1
    traffic, traffic_source = profiled_traffic(dom, kd["launches"]) if is_default else (None, None)
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(achieved / PEAK_HBM_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_us": round(avg_us, 2), "algorithmic_bytes_per_launch": int(kd["bytes"] / max(kd["launches"], 1)),
                "all_kernels": {k: {"total_ms": round(v["ms"], 3), "launches": int(v["launches"]),
                                    "GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 2)} for k, v in kernels.items()},
                "whole_step_GBps": round((si.extend_bytes + si.shade_bytes) / (ms_per_step * 1e-3) / 1e9, 2)}

    if rank == 0:
        out = {
            "metric": "Msamples/sec (whole node), cbox 256spp" if (os.path.basename(args.scene) == "cbox.xml" and spp == 256)
                      else f"Msamples/sec (whole node), {os.path.basename(args.scene)} {spp}spp",
            "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "scene file shipped with the reference (scenes/%s), sampleCount overridden to %d" % (os.path.relpath(args.scene, os.path.join(ROOT, "scenes")), spp),
            "config": {"workload": f"{os.path.basename(args.scene)} {w}x{h} @ {spp} spp, path integrator" + (" (Lambertian + area light, max_depth -1, rr_depth 5)" if os.path.basename(args.scene) == "cbox.xml" else f" (max_depth {hs.desc.options.max_depth}, rr_depth {hs.desc.options.rr_depth})"),
                       "samples_per_step": total_samples, "mean_bounce_iterations_K": round(k_mean, 4),
                       "rng": "pcg32, one stream per (pixel, sample), seed 0x853c49e6748fea9b",
                       "parallelism": f"tiles%{world}" if world > 1 else "1 GPU", "collective": "RCCL reduce(sum) of the float framebuffer" if world > 1 else "none"},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.scene)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
