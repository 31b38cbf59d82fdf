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
ROUND = "r03"              # profiles/<ROUND>_* hold this round's rocprofv3 evidence


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
        if not os.path.isfile(os.path.join(d, f)) or f.startswith("."):   # (sources only: no editor / test-runner droppings)
            continue
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


_COUNTERS = None


def _load_counters():
    global _COUNTERS
    if _COUNTERS is None:
        path = os.path.join(ROOT, "profiles", f"{ROUND}_counters.json")
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            _COUNTERS = (None, f"no profiles/{ROUND}_counters.json")
            return _COUNTERS
        if t.get("kernel_source_sha") != kernel_source_sha():
            _COUNTERS = (None, f"profiles/{ROUND}_counters.json was measured on other kernel sources (stale)")
        else:
            _COUNTERS = (t, None)
    return _COUNTERS


def profiled_counters(workload):
    """Hardware counters per kernel of `workload` ("cbox.xml@256", "sponza.xml@1024" ...) from this round's committed rocprofv3 PMC passes
    (profiles/<ROUND>_counters.json, written by tools/evidence_r03.sh: SQ sets for every bench workload, FETCH_SIZE / WRITE_SIZE in
    separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streams; totals over one render's launches).
    Returns (None, reason) when the file is missing or was measured on other kernel sources: a stale profile is never quoted."""
    t, why = _load_counters()
    if t is None:
        return None, why
    k = t.get("workloads", {}).get(workload)
    if not k:
        return None, f"{workload} not in profiles/{ROUND}_counters.json"
    # the extend kernels (k_extend, k_extend8) report under one name, as the library's per-kernel timers do
    out = {}
    for name, e in k.items():
        out["k_extend" if name.startswith("k_extend") else name] = e
    return out, None


def profiled_counters_source():
    t, _ = _load_counters()
    return t.get("source") if t else None


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
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE.json configs 3-5 after the headline loop")
    ap.add_argument("--config-steps", type=int, default=2, help="timed renders per extra configuration")
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

    ctx = lj.Context(local_rank)
    # An explicit, non-default stream: render, reduce and the next step's clear of `frame` are then all ordered on it (the
    # reduce is enqueued on RCCL's stream behind everything this stream holds at the call, and the stream waits for it
    # before anything later).  Stream 0 would mean "the context's own stream" to lj_render_device, which nothing in torch
    # orders against.
    stream = torch.cuda.Stream(device=dev)
    assert stream.cuda_stream != 0
    PEAK_GWINST = 1024 * 2.4 / 2   # 256 CUs x 4 SIMD-32, one wave64 instruction per 2 cycles, 2.4 GHz (MI355X_MICROARCH.md)

    def measure(scene_xml, spp, steps, warmup):
        """`steps` timed renders of one workload (barrier + synchronize on both sides, max over ranks), then one instrumented render
        (flags=1: HIP events around every launch, outside the timed region) for the per-kernel durations.  Returns the line's pieces."""
        hs = lj.parse_scene(scene_xml)
        scene = lj.Scene(ctx, hs)
        w, h = hs.width, hs.height
        total_samples = w * h * spp
        frame = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)

        def step():
            with torch.cuda.stream(stream):
                lj.render_device(scene, frame.data_ptr(), stream=stream.cuda_stream, spp=spp, rank=rank, world_size=world, pool_paths=args.pool)
                ljdist.reduce_framebuffer(frame, dst=0)

        for _ in range(warmup):
            step()
        ljdist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ljdist.barrier()
        torch.cuda.synchronize(dev)
        elapsed = ljdist.max_over_ranks(time.perf_counter() - t0, device=dev)
        ms_per_step = elapsed / steps * 1e3
        st = scene.stats()
        lj.render_device(scene, frame.data_ptr(), stream=stream.cuda_stream, spp=spp, rank=rank, world_size=world, pool_paths=args.pool, flags=1)
        torch.cuda.synchronize(dev)
        si = scene.stats()
        if scene.info.integrator == 6:   # volumetric path tracer: one launch per pass (k_volpath); timed as a whole
            kernels = {"k_volpath": {"ms": si.render_ms, "launches": max(int(si.wavefront_steps), 1), "bytes": si.samples * 12}}
        elif si.mega_launches > 0:   # a tiny scene: one fused persistent launch per pass (mega.hip), no path queue
            kernels = {"k_mega": {"ms": si.mega_ms, "launches": si.mega_launches, "bytes": si.mega_bytes}}
        else:
            kernels = {"k_extend": {"ms": si.extend_ms, "launches": si.extend_launches, "bytes": si.extend_bytes},
                       "k_shade": {"ms": si.shade_ms, "launches": si.shade_launches, "bytes": si.shade_bytes}}
        name = os.path.basename(scene_xml)
        return {"hs": hs, "name": name, "spp": spp, "w": w, "h": h, "total_samples": total_samples, "ms_per_step": ms_per_step,
                "value": total_samples * steps / elapsed / 1e6, "k_mean": st.bounce_iterations / max(st.samples, 1), "kernels": kernels, "si": si,
                "share": si.samples / max(total_samples, 1)}   # this rank's share of the frame's samples (1 at N = 1)

    def kernel_roofline(m, k, prof):
        """One kernel of one workload against the resource that bounds it.  k_shade streams the path queue: algorithmic HBM bytes per
        launch / live launch duration against the HBM peak.  k_extend (divergent BVH traversal), k_mega and k_volpath (paths in
        registers) move few bytes and are limited by vector-instruction issue: wave-instructions per second against the chip's issue
        peak, the instruction count from this round's committed PMC pass of the same workload (scaled by this rank's share of the
        samples for N > 1), the duration measured live here.  Their algorithmic HBM figure is reported beside it."""
        kd = m["kernels"][k]
        avg_us = kd["ms"] * 1e3 / max(kd["launches"], 1)
        gbs = (kd["bytes"] / max(kd["launches"], 1)) / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
        r = {"kernel": k, "bound": "hbm", "achieved": round(gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 5),
             "avg_launch_us": round(avg_us, 2), "launches": int(kd["launches"]), "algorithmic_bytes_per_launch": int(kd["bytes"] / max(kd["launches"], 1))}
        if k != "k_shade":
            r["bound"] = "valu"
            r["hbm_algorithmic"] = {"achieved": r["achieved"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": r["frac"]}
            if prof and prof.get("valu_wave_insts") and kd["ms"] > 0:
                insts = prof["valu_wave_insts"] * m["share"]
                a = insts / (kd["ms"] * 1e-3) / 1e9
                r.update({"achieved": round(a, 1), "peak": PEAK_GWINST, "unit": "G wave-instructions/s", "frac": round(a / PEAK_GWINST, 5),
                          "valu_active_lane_frac": prof.get("valu_active_lane_frac"), "valu_wave_insts_per_render": int(insts)})
                if prof.get("floor_wave_insts"):   # the issue floor: what the launch would take if every instruction it issues ran
                    # with all 64 lanes live, at the chip's issue peak, over what it takes — distance to a floor, where `frac` is a utilisation
                    r["floor_frac"] = round(prof["floor_wave_insts"] * m["share"] / (kd["ms"] * 1e-3) / 1e9 / PEAK_GWINST, 5)
                    r["floor_kind"] = prof.get("floor_kind")
            else:
                r["note"] = "VALU-bound kernel without a current PMC profile to take its instruction count from: the figures are its algorithmic HBM rate"
        if prof and "fetch_bytes" in prof and "write_bytes" in prof:
            r["traffic"] = int((prof["fetch_bytes"] + prof["write_bytes"]) * m["share"] / max(kd["launches"], 1))
        else:
            r["traffic"] = None
        return r

    def workload_rooflines(m):
        profs, why = profiled_counters(f"{m['name']}@{m['spp']}")
        out = {k: kernel_roofline(m, k, (profs or {}).get(k)) for k in m["kernels"]}
        dom = max(m["kernels"], key=lambda k: m["kernels"][k]["ms"])
        return dom, out, why

    m = measure(args.scene, args.spp, args.steps, args.warmup)
    dom, rl, why_not = workload_rooflines(m)
    roofline = dict(rl[dom])
    whole_gbs = (m["si"].extend_bytes + m["si"].shade_bytes + m["si"].mega_bytes) / (m["ms_per_step"] * 1e-3) / 1e9
    roofline.update({"traffic_source": (profiled_counters_source() if why_not is None else why_not),
                     "all_kernels": {k: {"total_ms": round(v["ms"], 3), "launches": int(v["launches"]), "bound": rl[k]["bound"], "frac": rl[k]["frac"]} for k, v in m["kernels"].items()},
                     "whole_step_GBps": round(whole_gbs, 2), "whole_step_frac": round(whole_gbs / PEAK_HBM_GBS, 5)})

    # ---- the other single-GPU configurations of BASELINE.json (3: disney_bsdf 256 spp, 4: veach_mi 512 spp, 5: sponza 1024 spp), a few
    # steps each, after the headline loop and outside its timed region; same timing discipline, reported under "configs"
    configs = []
    is_headline = os.path.basename(args.scene) == "cbox.xml" and args.spp == 256
    if is_headline and not args.no_configs:
        for rel, spp_c in (("disney_bsdf_test/disney_bsdf.xml", 256), ("veach_mi/mi.xml", 512), ("sponza/sponza.xml", 1024)):
            mc = measure(os.path.join(ROOT, "scenes", rel), spp_c, args.config_steps, 1)
            dom_c, rl_c, why_c = workload_rooflines(mc)
            e = {"workload": f"{mc['name']} {mc['w']}x{mc['h']} @ {spp_c} spp", "samples_per_step": mc["total_samples"], "steps": args.config_steps,
                 "ms_per_step": round(mc["ms_per_step"], 3), "value": round(mc["value"], 2), "unit": "Msamples/s",
                 "mean_bounce_iterations_K": round(mc["k_mean"], 4), "roofline": dict(rl_c[dom_c]),
                 "kernels": {k: {"total_ms": round(v["ms"], 3), "bound": rl_c[k]["bound"], "achieved": rl_c[k]["achieved"], "unit": rl_c[k]["unit"], "frac": rl_c[k]["frac"],
                                 "valu_active_lane_frac": rl_c[k].get("valu_active_lane_frac")} for k, v in mc["kernels"].items()}}
            if why_c:
                e["roofline"]["note"] = why_c
            configs.append(e)

    if rank == 0:
        name, spp, w, h, hs = m["name"], m["spp"], m["w"], m["h"], m["hs"]
        out = {
            "metric": "Msamples/sec (whole node), cbox 256spp" if (name == "cbox.xml" and spp == 256) else f"Msamples/sec (whole node), {name} {spp}spp",
            "value": round(m["value"], 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(m["ms_per_step"], 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "dtype_note": "float shading against the reference's double; parity bars (tests/test_gpu_parity.py): per-sample median rel. diff < 2e-6, image L2 <= 1e-2 at 16 spp",
            "data": "scene file shipped with the reference (scenes/%s), sampleCount overridden to %d" % (os.path.relpath(args.scene, os.path.join(ROOT, "scenes")), spp),
            "config": {"workload": f"{name} {w}x{h} @ {spp} spp, path integrator" + (" (Lambertian + area light, max_depth -1, rr_depth 5)" if name == "cbox.xml" else f" (max_depth {hs.desc.options.max_depth}, rr_depth {hs.desc.options.rr_depth})"),
                       "samples_per_step": m["total_samples"], "mean_bounce_iterations_K": round(m["k_mean"], 4),
                       "rng": "pcg32, one stream per (pixel, sample), seed 0x853c49e6748fea9b",
                       "parallelism": f"tiles%{world}" if world > 1 else "1 GPU", "collective": "RCCL reduce(sum) of the float framebuffer" if world > 1 else "none"},
            "roofline": roofline,
        }
        if configs:
            out["configs"] = configs
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.scene)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
