"""Build liblajolla_hip.so (product) — and, for the test infrastructure, oracle/_build/liblj_oracle.so.

    python -m lajolla_public_amd.build            # product library (hipcc --offload-arch=gfx950, cross-compiles without a GPU)
    python -m lajolla_public_amd.build --oracle   # also the CPU oracle (g++), test infrastructure only
    python -m lajolla_public_amd.build --twin     # also the host build of the device headers used by CPU-side tests

Objects are cached under build/ and rebuilt when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lajolla_public_amd", "csrc")
# LJ_VARIANT=name (developer A/B runs, tools/ab.sh): a second library beside the default one, built with
# LJ_EXTRA_HIPCC_FLAGS into its own object directory and picked up by the package when LJ_VARIANT is set at import
_VARIANT = os.environ.get("LJ_VARIANT", "")
BUILD = os.path.join(ROOT, "build" + ("_" + _VARIANT if _VARIANT else ""))
LIB = os.path.join(ROOT, "lajolla_public_amd", "liblajolla_hip" + ("_" + _VARIANT if _VARIANT else "") + ".so")
# LJ_SANITIZE=1 (tools/sanitize_cpu.sh): AddressSanitizer + UndefinedBehaviorSanitizer builds of the two CPU-side test libraries
# (the GPU pool offers no device sanitizer; the host twin compiles the very device headers, so it is the next best thing)
_SAN = bool(os.environ.get("LJ_SANITIZE"))
_SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"] if _SAN else []
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "liblj_oracle_asan.so" if _SAN else "liblj_oracle.so")
TWIN_LIB = os.path.join(ROOT, "tests", "twin", "_build", "libljtwin_asan.so" if _SAN else "libljtwin.so")

HOST_SOURCES = ["host/api_host.cpp", "host/scene_xml.cpp", "host/mesh_io.cpp", "host/image_io.cpp", "host/jpeg_decode.cpp", "host/exr_decode.cpp", "host/png_decode.cpp", "host/tga_bmp_decode.cpp", "host/flatten.cpp", "host/bvh.cpp"]
HIP_SOURCES = ["device/kernels.hip", "device/extend8.hip", "device/volpath.hip", "device/api_device.hip", "device/queries.hip", "device/mega.hip", "device/group.hip"]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _headers():
    out = [os.path.join(ROOT, "include", "lajolla_hip.h")]
    for d, _, files in os.walk(CSRC):
        out += [os.path.join(d, f) for f in files if f.endswith(".h")]
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: " + " ".join(cmd) + "\n" + r.stdout)
    if r.stdout.strip():
        sys.stderr.write(r.stdout)
    # the command line is part of an object's cache key (a sidecar next to it): a build with other flags — e.g.
    # LJ_EXTRA_HIPCC_FLAGS from tools/occ_sweep.sh — is rebuilt by the next default build instead of being silently reused
    if "-o" in cmd:
        with open(cmd[cmd.index("-o") + 1] + ".cmd", "w") as f:
            f.write(" ".join(cmd))


def _same_cmd(target, cmd):
    try:
        return open(target + ".cmd").read() == " ".join(cmd)
    except OSError:
        return False


def build_product(verbose=True):
    if os.environ.get("LJ_NO_REBUILD") and os.path.exists(LIB):   # developer A/B runs of a prebuilt LJ_VARIANT library on the GPU box
        return LIB
    # Extra compiler flags make a VARIANT of the library (developer A/B runs).  They may only go into a library of their own
    # (LJ_VARIANT=name -> liblajolla_hip_<name>.so) — never silently into the default one, where the next test run would load them.
    if os.environ.get("LJ_EXTRA_HIPCC_FLAGS") and not _VARIANT and os.environ.get("LJ_VARIANT_OK") != "1":
        raise RuntimeError("LJ_EXTRA_HIPCC_FLAGS is set without LJ_VARIANT: refusing to build the default library with variant flags "
                           "(set LJ_VARIANT=<name> for a side-by-side library, or LJ_VARIANT_OK=1 to override)")
    os.makedirs(BUILD, exist_ok=True)
    headers = _headers()
    jobs = []
    objs = []
    for src in HOST_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD, src.replace("/", "_") + ".o")
        objs.append(o)
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-Wall", "-Wno-unused-function", "-c", s, "-o", o]
        if _stale(o, [s] + headers) or not _same_cmd(o, cmd):
            jobs.append(cmd)
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD, src.replace("/", "_") + ".o")
        objs.append(o)
        # -fno-slp-vectorize: packed-fp32 pairs (v_pk_add/mul/fma_f32) cost more in register shuffling than they save
        # in these kernels (measured: -6 % cbox, -14 % veach_mi, -1..4 % elsewhere; 12 fewer VGPRs in the extend kernel)
        # (not built with -fno-hip-fp32-correctly-rounded-divide-sqrt: the 2.5-ulp divisions are 2 % faster on every scene, but the
        # volumetric tracker's long chains of accept / reject decisions then leave the oracle's path in most samples of a dense medium —
        # vol_cbox per-sample parity went from a median 2e-7 to 0.15 — and sponza's per-sample median crossed its 1e-4 bar: measured, dropped)
        cmd = [_hipcc(), "-std=c++17", "-O3", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
               "-fno-gpu-rdc", "-fno-slp-vectorize"] + os.environ.get("LJ_EXTRA_HIPCC_FLAGS", "").split() + ["-c", s, "-o", o]
        if _stale(o, [s] + headers) or not _same_cmd(o, cmd):
            jobs.append(cmd)
    if jobs:
        if verbose:
            print(f"[build] compiling {len(jobs)} translation unit(s) for {ARCH}", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(_run, jobs))
    if jobs or _stale(LIB, objs):
        _run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-lz"])
    return LIB


def _host_fma_flag():
    """-mfma when this CPU has it: the oracle's explicit fmaf() calls then compile to one instruction instead of a libm
    call (same result either way — fma is exactly specified)."""
    try:
        return ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
    except OSError:
        return []


def build_oracle(verbose=True):
    src = os.path.join(ROOT, "oracle", "lj_oracle.cpp")
    os.makedirs(os.path.dirname(ORACLE_LIB), exist_ok=True)
    if _stale(ORACLE_LIB, [src, os.path.join(ROOT, "include", "lajolla_hip.h")]):
        if verbose:
            print("[build] compiling the CPU oracle (test infrastructure)", file=sys.stderr)
        _run(["g++", "-std=c++17", "-O1" if _SAN else "-O2", "-ffp-contract=off"] + _SAN_FLAGS + _host_fma_flag() + ["-fPIC", "-shared", "-Wall", "-Wno-unused-function",
              "-o", ORACLE_LIB, src, "-lpthread"])
    return ORACLE_LIB


def build_twin(verbose=True):
    src = os.path.join(ROOT, "tests", "twin", "twin.cpp")
    if not os.path.exists(src):
        return None
    os.makedirs(os.path.dirname(TWIN_LIB), exist_ok=True)
    deps = [src] + _headers() + [os.path.join(CSRC, s) for s in ("host/flatten.cpp", "host/bvh.cpp")]
    if _stale(TWIN_LIB, deps):
        if verbose:
            print("[build] compiling the host twin of the device headers (CPU-side tests only)", file=sys.stderr)
        _run(["g++", "-std=c++17", "-O1" if _SAN else "-O2", "-ffp-contract=off"] + _SAN_FLAGS + _host_fma_flag() + ["-fPIC", "-shared", "-Wall", "-Wno-unused-function",
              "-o", TWIN_LIB, src, os.path.join(CSRC, "host/flatten.cpp"), os.path.join(CSRC, "host/bvh.cpp"), "-lpthread"])
    return TWIN_LIB


def build_reference_subset():
    """oracle/_ref from the reference's own sources — only where /root/reference exists (this container)."""
    script = os.path.join(ROOT, "oracle", "ref_build.sh")
    if os.path.isdir(os.environ.get("LJ_REFERENCE_ROOT", "/root/reference")):
        _run(["bash", script])


if __name__ == "__main__":
    build_product()
    if "--oracle" in sys.argv or "--all" in sys.argv:
        build_oracle()
    if "--twin" in sys.argv or "--all" in sys.argv:
        build_twin()
    print(LIB)
