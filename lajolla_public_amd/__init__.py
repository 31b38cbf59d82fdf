"""lajolla_public_amd — host-side mirror of lajolla's render interface over the MI355X (gfx950) hot path.

    hs    = parse_scene("scenes/cbox/cbox.xml")   # parse_scene(path, device)   src/parse_scene.h:9   (host, no GPU)
    ctx   = Context(device=0)                     # rtcNewDevice                src/main.cpp:30
    scene = Scene(ctx, hs)                        # Scene::Scene(...)           src/scene.cpp:3-53
    img   = render(scene)                         # Image3 render(const Scene&) src/render.h:9  -> (h, w, 3) float32

Everything goes through the C ABI of liblajolla_hip.so (include/lajolla_hip.h).  There is no CPU rendering path
in this package: if the library is missing, or no gfx950 device is present, the calls raise.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import _abi
from ._abi import LjRenderArgs, LjSceneDesc, LjStats, LjSceneInfo, LjRay, LjHit

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LJ_VARIANT: a developer build of the same library with other compile flags, see build.py; unset in every shipped path)
LIB_PATH = os.path.join(_HERE, "liblajolla_hip" + ("_" + os.environ["LJ_VARIANT"] if os.environ.get("LJ_VARIANT") else "") + ".so")
_lib = None


class LajollaError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[lj error {code}] {message}")
        self.code = code


def load_library():
    """dlopen liblajolla_hip.so and bind every symbol include/lajolla_hip.h declares.  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `python -m lajolla_public_amd.build`) from the repo root. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, restype, argtypes in _abi.SYMBOLS:
        fn = getattr(lib, name)  # AttributeError here == the library does not export what the header declares
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(rc):
    if rc != _abi.LJ_OK:
        raise LajollaError(rc, load_library().lj_last_error().decode("utf-8", "replace"))


class HostScene:
    """Result of the XML front end: owns an LjSceneDesc (the constructor arguments of the reference's Scene)."""

    def __init__(self, handle):
        self._h = handle
        self.desc_ptr = load_library().lj_host_scene_desc(handle)
        self.desc = self.desc_ptr.contents

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.lj_host_scene_free(self._h)
            self._h = None

    # -- convenience views (copies) used by tests and tools
    @property
    def width(self):
        return self.desc.camera.width

    @property
    def height(self):
        return self.desc.camera.height

    @property
    def spp(self):
        return self.desc.options.samples_per_pixel

    def positions(self):
        n = self.desc.n_vertices
        return np.ctypeslib.as_array(self.desc.positions, shape=(n, 3)).copy() if n else np.zeros((0, 3))

    def normals(self):
        n = self.desc.n_vertices
        return np.ctypeslib.as_array(self.desc.normals, shape=(n, 3)).copy() if n else np.zeros((0, 3))

    def uvs(self):
        n = self.desc.n_vertices
        return np.ctypeslib.as_array(self.desc.uvs, shape=(n, 2)).copy() if n else np.zeros((0, 2))

    def indices(self):
        n = self.desc.n_triangles
        return np.ctypeslib.as_array(self.desc.indices, shape=(n, 3)).copy() if n else np.zeros((0, 3), np.int32)


def parse_scene(path):
    """Mitsuba-0.x XML -> HostScene.  Host only (no GPU needed).  Mirrors parse_scene() (parse_scene.cpp:1134-1149)."""
    lib = load_library()
    h = C.c_void_p()
    _check(lib.lj_parse_scene(os.fsencode(path), C.byref(h)))
    return HostScene(h)


def write_image(path, rgb):
    """imwrite() (image.cpp:135-173): `rgb` is (h, w, 3); .pfm in the reference's top-down layout, or .exr (HALF)."""
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("write_image expects an (h, w, 3) array")
    _check(load_library().lj_image_write(os.fsencode(path), a.shape[1], a.shape[0], a.ctypes.data_as(C.c_void_p)))


def read_image(path, channels=3):
    """imread3 / imread1 (image.cpp:28-133): (h, w, channels) float32, y = 0 at the top."""
    lib = load_library()
    w, h, data = C.c_int32(), C.c_int32(), C.POINTER(C.c_float)()
    _check(lib.lj_image_read(os.fsencode(path), int(channels), C.byref(w), C.byref(h), C.byref(data)))
    try:
        return np.ctypeslib.as_array(data, shape=(h.value, w.value, int(channels))).copy()
    finally:
        lib.lj_image_free(data)


class Context:
    """One HIP device + stream + workspace."""

    def __init__(self, device=0):
        lib = load_library()
        self._h = C.c_void_p()
        self._scenes = 0        # live Scene objects on this context
        self._released = False  # __del__ has run while scenes were still alive: the last scene destroys the context
        _check(lib.lj_context_create(int(device), C.byref(self._h)))

    def _destroy(self):
        if getattr(self, "_h", None) and _lib is not None and not sys.is_finalizing():
            _lib.lj_context_destroy(self._h)
            self._h = None

    def __del__(self):
        # A scene keeps its context alive by reference, but when both die in one garbage-collected cycle (a failed test's
        # traceback holds them) Python may finalise the context first: it must then outlive the scenes at the C level.
        if getattr(self, "_scenes", 0) > 0:
            self._released = True
        else:
            self._destroy()


class Scene:
    """Device-resident scene: flattened BVH + sampling tables (the reference's Scene::Scene, scene.cpp:3-53)."""

    def __init__(self, ctx, host_scene_or_desc):
        lib = load_library()
        self._ctx = ctx  # keep the context alive
        self._h = C.c_void_p()
        desc_ptr = host_scene_or_desc.desc_ptr if isinstance(host_scene_or_desc, HostScene) else C.pointer(host_scene_or_desc)
        _check(lib.lj_scene_upload(ctx._h, desc_ptr, C.byref(self._h)))
        ctx._scenes += 1
        info = LjSceneInfo()
        _check(lib.lj_scene_info(self._h, C.byref(info)))
        self.info = info

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and not sys.is_finalizing():
            _lib.lj_scene_destroy(self._h)
            self._h = None
            self._ctx._scenes -= 1
            if self._ctx._released and self._ctx._scenes == 0:
                self._ctx._destroy()

    def stats(self):
        st = LjStats()
        _check(load_library().lj_get_stats(self._h, C.byref(st)))
        return st


def make_args(spp=0, max_depth=None, rank=0, world_size=1, crop=None, pool_paths=0, seed=0, flags=0):
    a = LjRenderArgs()
    a.spp = int(spp)
    a.max_depth = _abi.INT32_MIN if max_depth is None else int(max_depth)
    a.rng_mode = 0
    a.rank, a.world_size = int(rank), int(world_size)
    if crop is not None:
        a.crop_x0, a.crop_y0, a.crop_x1, a.crop_y1 = [int(v) for v in crop]
    a.pool_paths = int(pool_paths)
    a.flags = int(flags)
    a.seed = int(seed)
    return a


def render(scene, **kw):
    """Image3 render(const Scene&) (render.h:9): returns (h, w, 3) float32 radiance, y = 0 at the top."""
    args = make_args(**kw)
    out = np.empty((scene.info.height, scene.info.width, 3), np.float32)
    _check(load_library().lj_render(scene._h, C.byref(args), out.ctypes.data_as(C.c_void_p)))
    return out


def render_device(scene, device_ptr, stream=None, **kw):
    """Same, into caller-owned device memory (e.g. a torch tensor's data_ptr()) on a HIP stream; asynchronous."""
    args = make_args(**kw)
    _check(load_library().lj_render_device(scene._h, C.byref(args), C.c_void_p(int(device_ptr)), C.c_void_p(int(stream or 0))))


def render_samples(scene, crop, **kw):
    """Per-sample radiance over a crop window: (crop_h, crop_w, spp, 3) float32 — one path_tracing() value each."""
    args = make_args(crop=crop, **kw)
    spp = args.spp if args.spp > 0 else scene.info.spp
    x0, y0, x1, y1 = crop
    out = np.empty((y1 - y0, x1 - x0, spp, 3), np.float32)
    _check(load_library().lj_render_samples(scene._h, C.byref(args), out.ctypes.data_as(C.c_void_p)))
    return out


def _rays_array(org, dir, tnear, tfar):
    org = np.asarray(org, np.float32).reshape(-1, 3)
    n = org.shape[0]
    rays = np.zeros(n, dtype=np.dtype([("org", np.float32, 3), ("tnear", np.float32), ("dir", np.float32, 3), ("tfar", np.float32)]))
    rays["org"] = org
    rays["dir"] = np.asarray(dir, np.float32).reshape(-1, 3)
    rays["tnear"] = tnear
    rays["tfar"] = tfar
    return rays


HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32), ("shape_id", np.int32), ("prim_id", np.int32)])


def intersect(scene, org, dir, tnear=0.0, tfar=np.inf):
    """Batched intersect() (intersection.cpp:7-65, hit record only) on the device BVH."""
    rays = _rays_array(org, dir, tnear, tfar)
    hits = np.zeros(rays.shape[0], HIT_DTYPE)
    _check(load_library().lj_intersect(scene._h, rays.shape[0], rays.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p)))
    return hits


def occluded(scene, org, dir, tnear=0.0, tfar=np.inf):
    """Batched occluded() (intersection.cpp:67-85) on the device BVH."""
    rays = _rays_array(org, dir, tnear, tfar)
    occ = np.zeros(rays.shape[0], np.uint8)
    _check(load_library().lj_occluded(scene._h, rays.shape[0], rays.ctypes.data_as(C.c_void_p), occ.ctypes.data_as(C.c_void_p)))
    return occ.astype(bool)


# ------------------------------------------------------------------ per-object queries on the device (lajolla_hip.h)
# numpy structured arrays with the C structs' layouts in, the same out.  One query per GPU lane, answered by the device
# functions the shade kernels are built from; there is no host arithmetic behind any of these.
BSDF_QUERY, BSDF_RESULT = np.dtype(_abi.LjBsdfQuery), np.dtype(_abi.LjBsdfResult)
LIGHT_QUERY, LIGHT_RESULT = np.dtype(_abi.LjLightQuery), np.dtype(_abi.LjLightResult)
HIT_QUERY, HIT_RESULT = np.dtype(_abi.LjHitQuery), np.dtype(_abi.LjHitResult)
PRIMARY_QUERY, PRIMARY_RESULT = np.dtype(_abi.LjPrimaryQuery), np.dtype(_abi.LjPrimaryResult)
FILTER_QUERY = np.dtype(_abi.LjFilterQuery)
TEXTURE_QUERY = np.dtype(_abi.LjTextureQuery)
FRAME_QUERY, FRAME_RESULT = np.dtype(_abi.LjFrameQuery), np.dtype(_abi.LjFrameResult)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _q(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a.reshape(-1)


def shade_variant_count():
    return load_library().lj_shade_variant_count()


def scene_shade_variant(scene):
    return load_library().lj_scene_shade_variant(scene._h)


def bsdf_queries(scene, queries, variant=-1):
    """eval / pdf_sample_bsdf / sample_bsdf (material.h:126,147,161) of the device BSDF code, per query."""
    q = _q(queries, BSDF_QUERY)
    r = np.zeros(q.shape[0], BSDF_RESULT)
    _check(load_library().lj_bsdf_queries(scene._h, int(variant), q.shape[0], _vp(q), _vp(r)))
    return r


def light_queries(scene, queries, variant=-1):
    """sample_point_on_light / pdf_point_on_light / emission (light.h:46-67) + light_pmf, per query."""
    q = _q(queries, LIGHT_QUERY)
    r = np.zeros(q.shape[0], LIGHT_RESULT)
    _check(load_library().lj_light_queries(scene._h, int(variant), q.shape[0], _vp(q), _vp(r)))
    return r


def sample_light_queries(scene, u):
    u = np.ascontiguousarray(u, np.float32).reshape(-1)
    ids = np.zeros(u.shape[0], np.int32)
    _check(load_library().lj_sample_light_queries(scene._h, u.shape[0], _vp(u), _vp(ids)))
    return ids


def vertex_queries(scene, queries, variant=-1):
    """compute_shading_info + PathVertex assembly (intersection.cpp:38-62) for given hit records."""
    q = _q(queries, HIT_QUERY)
    r = np.zeros(q.shape[0], HIT_RESULT)
    _check(load_library().lj_vertex_queries(scene._h, int(variant), q.shape[0], _vp(q), _vp(r)))
    return r


def primary_ray_queries(scene, queries):
    q = _q(queries, PRIMARY_QUERY)
    r = np.zeros(q.shape[0], PRIMARY_RESULT)
    _check(load_library().lj_primary_ray_queries(scene._h, q.shape[0], _vp(q), _vp(r)))
    return r


def filter_queries(ctx, queries):
    q = _q(queries, FILTER_QUERY)
    r = np.zeros((q.shape[0], 2), np.float32)
    _check(load_library().lj_filter_queries(ctx._h, q.shape[0], _vp(q), _vp(r)))
    return r


def pcg32_queries(ctx, stream_ids, count, seed=0, reals=True):
    s = np.ascontiguousarray(stream_ids, np.uint64).reshape(-1)
    u = np.zeros((s.shape[0], count), np.uint32)
    f = np.zeros((s.shape[0], count), np.float32) if reals else None
    _check(load_library().lj_pcg32_queries(ctx._h, s.shape[0], _vp(s), int(seed), int(count), _vp(u), _vp(f) if reals else None))
    return u, f


def texture_queries(scene, queries):
    q = _q(queries, TEXTURE_QUERY)
    r = np.zeros((q.shape[0], 3), np.float32)
    _check(load_library().lj_texture_queries(scene._h, q.shape[0], _vp(q), _vp(r)))
    return r


def frame_queries(ctx, queries):
    q = _q(queries, FRAME_QUERY)
    r = np.zeros(q.shape[0], FRAME_RESULT)
    _check(load_library().lj_frame_queries(ctx._h, q.shape[0], _vp(q), _vp(r)))
    return r


# ------------------------------------------------------------------ device groups: one process, N devices (lajolla_hip.h)
class DeviceGroup:
    """N devices driven from this process; RCCL communicator inside when they are distinct and N >= 2."""

    def __init__(self, device_ids):
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        self._h = C.c_void_p()
        self._scenes = 0
        self._released = False
        _check(load_library().lj_group_create(len(device_ids), ids, C.byref(self._h)))
        self.size = load_library().lj_group_size(self._h)
        self.uses_rccl = bool(load_library().lj_group_uses_rccl(self._h))

    def _destroy(self):
        if getattr(self, "_h", None) and _lib is not None and not sys.is_finalizing():
            _lib.lj_group_destroy(self._h)
            self._h = None

    def __del__(self):
        if getattr(self, "_scenes", 0) > 0:
            self._released = True
        else:
            self._destroy()


class GroupScene:
    """The scene on every device of a group."""

    def __init__(self, group, host_scene_or_desc):
        self._group = group
        self._h = C.c_void_p()
        desc_ptr = host_scene_or_desc.desc_ptr if isinstance(host_scene_or_desc, HostScene) else C.pointer(host_scene_or_desc)
        _check(load_library().lj_group_scene_upload(group._h, desc_ptr, C.byref(self._h)))
        group._scenes += 1
        info = LjSceneInfo()
        _check(load_library().lj_scene_info(load_library().lj_group_scene_member(self._h, 0), C.byref(info)))
        self.info = info

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and not sys.is_finalizing():
            _lib.lj_group_scene_destroy(self._h)
            self._h = None
            self._group._scenes -= 1
            if self._group._released and self._group._scenes == 0:
                self._group._destroy()

    def stats(self):
        st = LjStats()
        _check(load_library().lj_group_get_stats(self._h, C.byref(st)))
        return st


def render_group(group_scene, **kw):
    """render() across a device group: tiles sharded t % N, one sum-reduce onto device 0 -> (h, w, 3) float32."""
    args = make_args(**kw)
    out = np.empty((group_scene.info.height, group_scene.info.width, 3), np.float32)
    _check(load_library().lj_group_render(group_scene._h, C.byref(args), out.ctypes.data_as(C.c_void_p)))
    return out
