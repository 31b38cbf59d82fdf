#!/usr/bin/env python3
"""bench.py — Msamples/s of the path-tracing hot path on MI355X.

One "step" = one complete render of BASELINE.json config 2: scenes/cbox/cbox.xml at 512x512, 256 spp
(67,108,864 camera samples; Lambertian + area light; max_depth -1, Russian roulette from depth 5), scene already
resident in HBM, timed from the first kernel to the resolved framebuffer (and, for N > 1, the RCCL reduce onto
rank 0) — the region the reference times at main.cpp:40-42.  N > 1: one process per GPU, the image's 16x16 tiles are
dealt round-robin to the ranks, each rank renders its tiles at full spp, one sum-reduce.

    python bench.py                      # 1 GPU
    python bench.py --gpus 8             # starts 8 ranks itself (children, before this process touches a GPU)
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8    # or under an outside launcher

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
COPY_HBM_GBS = 6290.0      # what a float4 copy kernel achieves on it (same guide): the practical ceiling
ROUND = "r02"              # profiles/<ROUND>_* hold this round's rocprofv3 evidence


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


def kernel_source_sha():
    """Hash of the device sources: a committed counter profile is only quoted while the kernels it measured are the
    kernels being timed."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "lajolla_public_amd", "csrc", "device")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def profiled_counters(kernel):
    """Hardware counters of `kernel` from this round's committed rocprofv3 PMC passes of the same workload
    (profiles/<ROUND>_counters.json, written by tools/evidence_r02.sh: FETCH_SIZE, WRITE_SIZE and two SQ sets in separate
    runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streams; totals over one render's launches).
    Returns (None, reason) when the file is missing or was measured on other kernel sources: a stale profile is never quoted."""
    path = os.path.join(ROOT, "profiles", f"{ROUND}_counters.json")
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None, f"no profiles/{ROUND}_counters.json"
    if t.get("kernel_source_sha") != kernel_source_sha():
        return None, f"profiles/{ROUND}_counters.json was measured on other kernel sources (stale)"
    k = t.get("kernels", {}).get(kernel)
    if not k or "fetch_bytes" not in k or "write_bytes" not in k:
        return None, f"{kernel} not in profiles/{ROUND}_counters.json"
    out = dict(k)
    out["traffic"] = int((k["fetch_bytes"] + k["write_bytes"]) / max(k["launches"], 1))
    out["source"] = t.get("source")
    return out, None


def cpu_baseline(scene_xml, seconds_budget=12.0):
    """The CPU restatement of the reference's parallel.cpp tile path (oracle/, kind `port`) on a bounded sample of the same
    workload: the same scene at a reduced spp chosen to take roughly `seconds_budget` on all host cores, then the same
    again on one thread.  The port traces through its own median-split BVH (for every scene size), not Embree."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lajolla_public_amd as lj
    from helpers import Oracle
    hs = lj.parse_scene(scene_xml)
    o = Oracle(hs)
    o.use_bvh(True)
    cores = host_cores()
    rc, _, _, st = o.render(spp=1, rng_mode=1, threads=cores)  # probe
    rate = st.samples / max(st.seconds, 1e-9)
    spp = int(max(1, min(256, round(seconds_budget * rate / (hs.width * hs.height)))))
    rc, _, _, st = o.render(spp=spp, rng_mode=1, threads=cores)
    spp1 = int(max(1, min(spp, round(spp / cores * 0.6))))
    rc, _, _, st1 = o.render(spp=spp1, rng_mode=1, threads=1)
    name = os.path.basename(scene_xml)
    return {"value": round(st.samples / st.seconds / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "one_thread_value": round(st1.samples / st1.seconds / 1e6, 4),
            "sample": f"{name} {hs.width}x{hs.height} at {spp} spp ({st.samples} samples, {st.seconds:.1f} s) on {cores} threads pulling 16x16 tiles from a "
                      f"shared counter; {spp1} spp ({st1.seconds:.1f} s) on 1 thread; reference RNG schedule (one pcg32 stream per tile), double precision",
            "note": "oracle/lj_oracle.cpp (our restatement, scalar, median-split BVH) — not the reference's Embree build, which cannot be "
                    "linked here (libembree3.so.3 is absent); Embree's SIMD traversal would be faster than this"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """--gpus N with no launcher around us: start the N ranks as CHILD processes (torch.distributed.run, one per GPU) before
    this process has made any GPU call, relay their output and return their exit code.  Never exec from here."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def dry_launch(args, rank, world):
    """--dry-launch: the launch / rendezvous / reduce / report plumbing of an N-rank run on CPU (gloo), no rendering: every rank
    fills the pixels of ITS tiles with a known pattern, the frames are sum-reduced onto rank 0 exactly as the rendered ones
    are, and rank 0 checks that every pixel arrived exactly once.  Used by tests/test_bench_launch.py."""
    import numpy as np
    import torch
    from lajolla_public_amd import dist as ljdist
    if world > 1:
        ljdist.init_process_group("gloo")
    w, h = 512, 512
    want = np.arange(h * w * 3, dtype=np.float32).reshape(h, w, 3) + 1.0
    frame = torch.from_numpy(want * ljdist.tile_owner_mask(w, h, rank, world)[..., None].astype(np.float32))
    ljdist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        part = frame.clone()
        ljdist.reduce_framebuffer(part, dst=0)
    ljdist.barrier()
    elapsed = ljdist.max_over_ranks(time.perf_counter() - t0)
    ok = True
    if rank == 0:
        ok = bool(np.array_equal(part.numpy(), want))
        print(json.dumps({"metric": "dry launch (no rendering)", "value": None, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "dry_launch": True, "reduce_exact": ok,
                          "config": {"workload": "tile-ownership pattern 512x512", "parallelism": f"tiles%{world}", "collective": "gloo reduce(sum) of the float framebuffer"}}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0 if ok else 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "cbox", "cbox.xml"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pool", type=int, default=0)
    ap.add_argument("--dry-launch", action="store_true", help="CPU/gloo rehearsal of the N-rank launch, reduce and report; renders nothing")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))   # nothing above has touched torch or HIP

    from lajolla_public_amd import dist as ljdist
    rank, world, local_rank = ljdist.env_rank_world()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.dry_launch:
        sys.exit(dry_launch(args, rank, world))

    import torch
    import lajolla_public_amd as lj
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
    # An explicit, non-default stream: render, reduce and the next step's clear of `frame` are then all ordered on it (the
    # reduce is enqueued on RCCL's stream behind everything this stream holds at the call, and the stream waits for it
    # before anything later).  Stream 0 would mean "the context's own stream" to lj_render_device, which nothing in torch
    # orders against.
    stream = torch.cuda.Stream(device=dev)
    assert stream.cuda_stream != 0

    def step():
        with torch.cuda.stream(stream):
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
    if scene.info.integrator == 6:   # volumetric path tracer: one launch per pass, a lane walks a whole path (k_volpath); timed as a whole
        kernels = {"k_volpath": {"ms": si.render_ms, "launches": max(int(si.wavefront_steps), 1), "bytes": si.samples * 12}}
    elif si.mega_launches > 0:   # a tiny scene: one fused persistent launch per pass (mega.hip), no path queue
        kernels = {"k_mega": {"ms": si.mega_ms, "launches": si.mega_launches, "bytes": si.mega_bytes}}
    else:
        kernels = {
            "k_extend": {"ms": si.extend_ms, "launches": si.extend_launches, "bytes": si.extend_bytes},
            "k_shade": {"ms": si.shade_ms, "launches": si.shade_launches, "bytes": si.shade_bytes},
        }
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    kd = kernels[dom]
    avg_us = kd["ms"] * 1e3 / max(kd["launches"], 1)
    achieved = (kd["bytes"] / max(kd["launches"], 1)) / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
    is_default = os.path.basename(args.scene) == "cbox.xml" and spp == 256 and world == 1
    prof, why_not = profiled_counters(dom) if is_default else (None, "only profiled for the default workload")
    whole_gbs = (si.extend_bytes + si.shade_bytes + si.mega_bytes) / (ms_per_step * 1e-3) / 1e9
    # `bound`: the shade kernel streams the queue (HBM).  The extend kernel is divergent BVH traversal whose queue traffic is a
    # fraction of its time, and k_mega keeps the path state in registers (its only HBM traffic is the finished radiance): both are
    # limited by vector-instruction issue, so their roofline is wave-instructions per second against the chip's issue peak
    # (256 CUs x 4 SIMD-32, one wave64 instruction per 2 cycles, 2.4 GHz: MI355X_MICROARCH.md) — with the instruction count taken
    # from this round's committed PMC pass of the same workload and the duration measured live here.  Their algorithmic HBM
    # figure is still reported beside it (hbm_*).
    PEAK_GWINST = 1024 * 2.4 / 2
    hbm = {"achieved": round(achieved, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 5),
           "frac_of_measured_copy": round(achieved / COPY_HBM_GBS, 5)}
    roofline = {"bound": "hbm", "kernel": dom}
    roofline.update(hbm)
    if dom != "k_shade":
        roofline["bound"] = "valu"
        if prof and prof.get("valu_wave_insts"):
            insts_per_launch = prof["valu_wave_insts"] / max(prof["launches"], 1)
            a = insts_per_launch / (avg_us * 1e-6) / 1e9
            roofline.update({"achieved": round(a, 1), "peak": PEAK_GWINST, "unit": "G wave-instructions/s", "frac": round(a / PEAK_GWINST, 5),
                             "valu_active_lane_frac": prof.get("valu_active_lane_frac"), "valu_wave_insts_per_launch": int(insts_per_launch)})
            roofline.pop("frac_of_measured_copy", None)
            roofline["hbm_algorithmic"] = hbm
        else:
            roofline["note"] = "VALU-bound kernel, but no current PMC profile to take its instruction count from (%s): the figures are its algorithmic HBM rate" % why_not
    roofline.update({"traffic": prof["traffic"] if prof else None, "traffic_source": prof["source"] if prof else why_not,
                     "avg_launch_us": round(avg_us, 2), "algorithmic_bytes_per_launch": int(kd["bytes"] / max(kd["launches"], 1)),
                     "all_kernels": {k: {"total_ms": round(v["ms"], 3), "launches": int(v["launches"]),
                                         "GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 2)} for k, v in kernels.items()},
                     "whole_step_GBps": round(whole_gbs, 2), "whole_step_frac": round(whole_gbs / PEAK_HBM_GBS, 5),
                     "whole_step_frac_of_measured_copy": round(whole_gbs / COPY_HBM_GBS, 5)})

    if rank == 0:
        name = os.path.basename(args.scene)
        out = {
            "metric": "Msamples/sec (whole node), cbox 256spp" if (name == "cbox.xml" and spp == 256) else f"Msamples/sec (whole node), {name} {spp}spp",
            "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "dtype_note": "float shading against the reference's double; parity bars (tests/test_gpu_parity.py): per-sample median rel. diff < 2e-6, image L2 <= 1e-2 at 16 spp",
            "data": "scene file shipped with the reference (scenes/%s), sampleCount overridden to %d" % (os.path.relpath(args.scene, os.path.join(ROOT, "scenes")), spp),
            "config": {"workload": f"{name} {w}x{h} @ {spp} spp, path integrator" + (" (Lambertian + area light, max_depth -1, rr_depth 5)" if name == "cbox.xml" else f" (max_depth {hs.desc.options.max_depth}, rr_depth {hs.desc.options.rr_depth})"),
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
