"""Test-side bindings: the CPU oracle (oracle/_build/liblj_oracle.so), the host twin of the device headers
(tests/twin/_build/libljtwin.so), golden-vector loading.  Nothing here is imported by the product."""
import ctypes as C
import json
import os

import numpy as np

import lajolla_public_amd as lj
from lajolla_public_amd import _abi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(ROOT, "scenes")

_dp = C.POINTER(C.c_double)


def golden(name):
    def fix(o):
        if isinstance(o, str) and o in ("nan", "inf", "-inf"):
            return float(o)
        if isinstance(o, list):
            return [fix(x) for x in o]
        if isinstance(o, dict):
            return {k: fix(v) for k, v in o.items()}
        return o
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return fix(json.load(f))


def darr(x):
    return np.ascontiguousarray(np.asarray(x, np.float64))


def dptr(a):
    return a.ctypes.data_as(_dp)


class OracleRenderArgs(C.Structure):
    _fields_ = [("spp", C.c_int32), ("rng_mode", C.c_int32), ("n_threads", C.c_int32), ("use_max_depth", C.c_int32),
                ("max_depth", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
                ("crop_x0", C.c_int32), ("crop_y0", C.c_int32), ("crop_x1", C.c_int32), ("crop_y1", C.c_int32),
                ("seed", C.c_uint64)]


class OracleStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("bounces", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("seconds", C.c_double), ("status", C.c_int32)]


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        _oracle = C.CDLL(build.build_oracle(verbose=False))
        _oracle.oracle_scene_create.restype = C.c_void_p
        _oracle.oracle_scene_create.argtypes = [C.POINTER(_abi.LjSceneDesc)]
        _oracle.oracle_scene_free.argtypes = [C.c_void_p]
        _oracle.oracle_mesh_total_area.restype = C.c_double
        _oracle.oracle_mesh_total_area.argtypes = [C.c_void_p, C.c_int]
        _oracle.oracle_sample_light.argtypes = [C.c_void_p, C.c_double]
        _oracle.oracle_render.argtypes = [C.c_void_p, C.POINTER(OracleRenderArgs), _dp, _dp, C.POINTER(OracleStats)]
    return _oracle


def oracle_phase(phase_kind, g, dir_in, dir_out, uv):
    ev = C.c_double()
    smp = np.zeros(3)
    oracle_lib().oracle_phase(C.c_int(phase_kind), C.c_double(g), dptr(darr(dir_in)), dptr(darr(dir_out)), dptr(darr(uv)), C.byref(ev), dptr(smp))
    return ev.value, smp


class Oracle:
    """CPU restatement of the reference, built from an LjSceneDesc (keeps the HostScene alive)."""

    def __init__(self, host_scene):
        self.hs = host_scene
        self.lib = oracle_lib()
        self.h = C.c_void_p(self.lib.oracle_scene_create(host_scene.desc_ptr))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_scene_free(self.h)
            self.h = None

    def use_bvh(self, on):
        self.lib.oracle_scene_use_bvh(self.h, C.c_int(1 if on else 0))

    def tables(self):
        n = self.hs.desc.n_lights
        br, eps = C.c_double(), C.c_double()
        bc = np.zeros(3)
        pmf, cdf, power = np.zeros(max(n, 1)), np.zeros(n + 1), np.zeros(max(n, 1))
        self.lib.oracle_scene_tables(self.h, C.byref(br), dptr(bc), C.byref(eps), dptr(pmf), dptr(cdf), dptr(power))
        return dict(bounds_radius=br.value, bounds_center=bc, shadow_epsilon=eps.value, light_pmf=pmf[:n], light_cdf=cdf, light_power=power[:n])

    def sample_primary(self, screen_pos):
        sp = darr(screen_pos).reshape(-1, 2)
        org, d = np.zeros((len(sp), 3)), np.zeros((len(sp), 3))
        self.lib.oracle_sample_primary(self.h, C.c_int(len(sp)), dptr(sp), dptr(org), dptr(d))
        return org, d

    def sample_light(self, u):
        return self.lib.oracle_sample_light(self.h, C.c_double(u))

    def light_sample(self, light_id, ref, uv, w, view_dir, footprint):
        pos, nrm, em = np.zeros(3), np.zeros(3), np.zeros(3)
        pdf = C.c_double()
        self.lib.oracle_light_sample(self.h, C.c_int(light_id), dptr(darr(ref)), dptr(darr(uv)), C.c_double(w), dptr(darr(view_dir)),
                                     C.c_double(footprint), dptr(pos), dptr(nrm), C.byref(pdf), dptr(em))
        return pos, nrm, pdf.value, em

    def make_vertex(self, org, d, rd_radius, rd_spread, shape_id, prim_id, t, u, v, Ng):
        out, em = np.zeros(22), np.zeros(3)
        mid = C.c_int()
        self.lib.oracle_make_vertex(self.h, dptr(darr(org)), dptr(darr(d)), C.c_double(rd_radius), C.c_double(rd_spread), C.c_int(shape_id),
                                    C.c_int(prim_id), C.c_float(t), C.c_float(u), C.c_float(v), dptr(darr(Ng)), dptr(out), C.byref(mid), dptr(em))
        return out, mid.value, em

    def bsdf(self, material, vertex22, dir_in, dir_out, rnd_uv, rnd_w, to_view=0):
        ev, sd = np.zeros(3), np.zeros(3)
        pdf, eta, rough = C.c_double(), C.c_double(), C.c_double()
        valid = C.c_int()
        rc = self.lib.oracle_bsdf(self.h, C.byref(material), dptr(darr(vertex22)), dptr(darr(dir_in)), dptr(darr(dir_out)), dptr(darr(rnd_uv)),
                                  C.c_double(rnd_w), C.c_int(to_view), dptr(ev), C.byref(pdf), C.byref(valid), dptr(sd), C.byref(eta), C.byref(rough))
        return rc, ev, pdf.value, valid.value, sd, eta.value, rough.value

    def medium_point(self, medium_id, p):
        ss, sa = np.zeros(3), np.zeros(3)
        self.lib.oracle_medium_point(self.h, C.c_int(medium_id), dptr(darr(p)), dptr(ss), dptr(sa))
        return ss, sa

    def medium_majorant(self, medium_id, org, d, tfar):
        out = np.zeros(3)
        self.lib.oracle_medium_majorant(self.h, C.c_int(medium_id), dptr(darr(org)), dptr(darr(d)), C.c_double(tfar), dptr(out))
        return out

    def intersect(self, rays):
        hits = np.zeros(rays.shape[0], lj.HIT_DTYPE)
        self.lib.oracle_intersect(self.h, C.c_int64(rays.shape[0]), rays.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p))
        return hits

    def occluded(self, rays):
        occ = np.zeros(rays.shape[0], np.uint8)
        self.lib.oracle_occluded(self.h, C.c_int64(rays.shape[0]), rays.ctypes.data_as(C.c_void_p), occ.ctypes.data_as(C.c_void_p))
        return occ.astype(bool)

    def render(self, spp=0, rng_mode=0, threads=0, max_depth=None, crop=None, rank=0, world_size=1, per_sample=False, seed=0):
        a = OracleRenderArgs()
        a.spp, a.rng_mode, a.n_threads = spp, rng_mode, threads
        a.use_max_depth, a.max_depth = (0, 0) if max_depth is None else (1, max_depth)
        a.rank, a.world_size = rank, world_size
        if crop:
            a.crop_x0, a.crop_y0, a.crop_x1, a.crop_y1 = crop
        a.seed = seed
        w, h = self.hs.width, self.hs.height
        spp_eff = spp if spp > 0 else self.hs.spp
        rgb = np.zeros((h, w, 3))
        ps = None
        if per_sample:
            x0, y0, x1, y1 = crop
            ps = np.zeros((y1 - y0, x1 - x0, spp_eff, 3))
        st = OracleStats()
        rc = self.lib.oracle_render(self.h, C.byref(a), dptr(rgb), dptr(ps) if ps is not None else None, C.byref(st))
        return rc, rgb, ps, st


_twin = None


def twin_lib():
    global _twin
    if _twin is None:
        _twin = C.CDLL(build.build_twin(verbose=False))
        _twin.twin_create.restype = C.c_void_p
        _twin.twin_create.argtypes = [C.POINTER(_abi.LjSceneDesc), C.c_char_p, C.c_int]
        _twin.twin_free.argtypes = [C.c_void_p]
    return _twin


class Twin:
    """Host build of the device headers (float), for CPU-side debugging of the kernel logic."""

    def __init__(self, host_scene):
        self.hs = host_scene
        self.lib = twin_lib()
        err = C.create_string_buffer(512)
        self.h = C.c_void_p(self.lib.twin_create(host_scene.desc_ptr, err, 512))
        if not self.h:
            raise RuntimeError(err.value.decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.twin_free(self.h)
            self.h = None

    def tables(self):
        n = self.hs.desc.n_lights
        br, eps = C.c_double(), C.c_double()
        bc = np.zeros(3)
        pmf, cdf, power = np.zeros(max(n, 1)), np.zeros(n + 1), np.zeros(max(n, 1))
        nn, depth = C.c_int(), C.c_int()
        self.lib.twin_tables(self.h, C.byref(br), dptr(bc), C.byref(eps), dptr(pmf), dptr(cdf), dptr(power), C.byref(nn), C.byref(depth))
        return dict(bounds_radius=br.value, bounds_center=bc, shadow_epsilon=eps.value, light_pmf=pmf[:n], light_cdf=cdf, light_power=power[:n],
                    n_nodes=nn.value, bvh_depth=depth.value)

    def render_samples(self, crop, spp, max_depth=None, threads=0, seed=0):
        x0, y0, x1, y1 = crop
        out = np.zeros((y1 - y0, x1 - x0, spp, 3), np.float32)
        b = C.c_ulonglong()
        self.lib.twin_render_samples(self.h, C.c_int(spp), C.c_int(0 if max_depth is None else max_depth), C.c_int(0 if max_depth is None else 1),
                                     C.c_uint64(seed), C.c_int(x0), C.c_int(y0), C.c_int(x1), C.c_int(y1), C.c_int(threads),
                                     out.ctypes.data_as(C.c_void_p), C.byref(b))
        return out, b.value

    def aux(self, integrator, crop):
        x0, y0, x1, y1 = crop
        out = np.zeros((y1 - y0, x1 - x0, 3), np.float32)
        self.lib.twin_aux(self.h, C.c_int(integrator), C.c_int(x0), C.c_int(y0), C.c_int(x1), C.c_int(y1), out.ctypes.data_as(C.c_void_p))
        return out

    def intersect(self, rays):
        hits = np.zeros(rays.shape[0], lj.HIT_DTYPE)
        self.lib.twin_intersect(self.h, C.c_int64(rays.shape[0]), rays.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p))
        return hits

    def intersect8(self, rays, work=None):
        """The same query over the scene's BVH8 (DNode8); work: uint64[2] accumulating node steps and primitive tests."""
        hits = np.zeros(rays.shape[0], lj.HIT_DTYPE)
        self.lib.twin_intersect8(self.h, C.c_int64(rays.shape[0]), rays.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p),
                                 work.ctypes.data_as(C.c_void_p) if work is not None else None)
        return hits

    def occluded8(self, rays):
        occ = np.zeros(rays.shape[0], np.uint8)
        self.lib.twin_occluded8(self.h, C.c_int64(rays.shape[0]), rays.ctypes.data_as(C.c_void_p), occ.ctypes.data_as(C.c_void_p))
        return occ.astype(bool)

    def intersect_work(self, rays):
        work = np.zeros(2, np.uint64)
        self.lib.twin_intersect_work(self.h, C.c_int64(rays.shape[0]), rays.ctypes.data_as(C.c_void_p), work.ctypes.data_as(C.c_void_p))
        return work

    def bvh8_info(self):
        out = np.zeros(4, np.int64)
        self.lib.twin_bvh8_info(self.h, out.ctypes.data_as(C.c_void_p))
        return dict(nodes=int(out[0]), depth=int(out[1]), filled_slots=int(out[2]), leaf_slots=int(out[3]))

    def occluded(self, rays):
        occ = np.zeros(rays.shape[0], np.uint8)
        self.lib.twin_occluded(self.h, C.c_int64(rays.shape[0]), rays.ctypes.data_as(C.c_void_p), occ.ctypes.data_as(C.c_void_p))
        return occ.astype(bool)


def scene_path(name):
    return {"cbox": os.path.join(SCENES, "cbox", "cbox.xml"), "veach_mi": os.path.join(SCENES, "veach_mi", "mi.xml"),
            "disney_bsdf": os.path.join(SCENES, "disney_bsdf_test", "disney_bsdf.xml"), "sponza": os.path.join(SCENES, "sponza", "sponza.xml")}[name]


def random_rays(hs, n, seed, oracle=None):
    """Rays aimed into the scene: random origins inside the bounds sphere, random directions; plus camera rays."""
    rng = np.random.default_rng(seed)
    tb = (oracle or Oracle(hs)).tables()
    c, r = tb["bounds_center"], tb["bounds_radius"]
    org = c + (rng.random((n, 3)) * 2 - 1) * r * 0.6
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return lj._rays_array(org, d, 0.0, np.inf)


def material_struct(m):
    """material dict (the layout of tests/golden/materials.json) -> LjMaterial"""
    out = _abi.LjMaterial()
    out.kind = _abi.MATERIAL_KINDS.index(m["kind"])
    slots = _abi.MATERIAL_SLOTS[m["kind"]]
    out.n_tex = len(slots)
    out.eta = m.get("eta", 0.0)
    tk = {"constant": 0, "image": 1, "checkerboard": 2}
    for i, sname in enumerate(slots):
        t = m[sname]
        tex = out.tex[i]
        tex.kind = tk[t["kind"]]
        tex.texture_id = t.get("texture_id", -1)
        v = t.get("value", t.get("color0", 0.0))
        v = [v] * 3 if not isinstance(v, list) else v
        c1 = t.get("color1", 0.0)
        c1 = [c1] * 3 if not isinstance(c1, list) else c1
        for k in range(3):
            tex.value[k], tex.color1[k] = v[k], c1[k]
        tex.uscale, tex.vscale = t.get("uscale", 1.0), t.get("vscale", 1.0)
        tex.uoffset, tex.voffset = t.get("uoffset", 0.0), t.get("voffset", 0.0)
    return out




def const(v):
    return {"kind": "constant", "value": v}


def set_material(hs, index, mdict):
    """Overwrite material `index` of a parsed scene (the description is plain memory owned by the HostScene)."""
    import ctypes
    m = material_struct(mdict)
    ctypes.memmove(ctypes.addressof(hs.desc.materials[index]), ctypes.addressof(m), ctypes.sizeof(m))


# ------------------------------------------------------------------ per-object queries: one interface, two executors
class TwinQueries:
    """lj_*_queries answered by the HOST build of the device headers (tests/twin): the CPU suite's stand-in for the GPU, so
    that the comparisons of tests/test_device_kats.py are themselves tested before a GPU minute is spent."""
    name = "twin"

    def __init__(self, hs=None):
        self.lib = twin_lib()
        self.tw = Twin(hs) if hs is not None else None

    def variants(self):
        return [v for v in range(lj.shade_variant_count()) if self.lib.twin_variant_covers(self.tw.h, C.c_int(v))]

    def _v(self, variant):
        return C.c_int(self.lib.twin_shade_variant(self.tw.h) if variant < 0 else variant)

    def bsdf(self, q, variant=-1):
        q = np.ascontiguousarray(q, lj.BSDF_QUERY).reshape(-1)
        r = np.zeros(len(q), lj.BSDF_RESULT)
        self.lib.twin_bsdf_queries(self.tw.h, self._v(variant), C.c_int64(len(q)), q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
        return r

    def light(self, q, variant=-1):
        q = np.ascontiguousarray(q, lj.LIGHT_QUERY).reshape(-1)
        r = np.zeros(len(q), lj.LIGHT_RESULT)
        self.lib.twin_light_queries(self.tw.h, self._v(variant), C.c_int64(len(q)), q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
        return r

    def sample_light(self, u):
        u = np.ascontiguousarray(u, np.float32).reshape(-1)
        ids = np.zeros(len(u), np.int32)
        self.lib.twin_sample_light_queries(self.tw.h, C.c_int64(len(u)), u.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p))
        return ids

    def vertex(self, q, variant=-1):
        q = np.ascontiguousarray(q, lj.HIT_QUERY).reshape(-1)
        r = np.zeros(len(q), lj.HIT_RESULT)
        self.lib.twin_vertex_queries(self.tw.h, self._v(variant), C.c_int64(len(q)), q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
        return r

    def primary(self, q):
        q = np.ascontiguousarray(q, lj.PRIMARY_QUERY).reshape(-1)
        r = np.zeros(len(q), lj.PRIMARY_RESULT)
        self.lib.twin_primary_ray_queries(self.tw.h, C.c_int64(len(q)), q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
        return r

    def filter(self, q):
        q = np.ascontiguousarray(q, lj.FILTER_QUERY).reshape(-1)
        r = np.zeros((len(q), 2), np.float32)
        self.lib.twin_filter_queries(C.c_int64(len(q)), q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
        return r

    def pcg32(self, streams, count, seed=0):
        s = np.ascontiguousarray(streams, np.uint64).reshape(-1)
        u, f = np.zeros((len(s), count), np.uint32), np.zeros((len(s), count), np.float32)
        self.lib.twin_pcg32_queries(C.c_int64(len(s)), s.ctypes.data_as(C.c_void_p), C.c_uint64(seed), C.c_int(count), u.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p))
        return u, f

    def texture(self, q):
        q = np.ascontiguousarray(q, lj.TEXTURE_QUERY).reshape(-1)
        r = np.zeros((len(q), 3), np.float32)
        self.lib.twin_texture_queries(self.tw.h, C.c_int64(len(q)), q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
        return r

    def frame(self, q):
        q = np.ascontiguousarray(q, lj.FRAME_QUERY).reshape(-1)
        r = np.zeros(len(q), lj.FRAME_RESULT)
        self.lib.twin_frame_queries(C.c_int64(len(q)), q.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p))
        return r


class GpuQueries:
    """The same interface through the C ABI of liblajolla_hip.so on cuda:0 (queries.hip)."""
    name = "gpu"
    _ctx = None

    def __init__(self, hs=None):
        if GpuQueries._ctx is None:
            GpuQueries._ctx = lj.Context(0)
        self.ctx = GpuQueries._ctx
        self.scene = lj.Scene(self.ctx, hs) if hs is not None else None

    def variants(self):
        out = []
        for v in range(lj.shade_variant_count()):
            try:
                lj.bsdf_queries(self.scene, np.zeros(0, lj.BSDF_QUERY), v)
                out.append(v)
            except lj.LajollaError:
                pass
        return out

    def bsdf(self, q, variant=-1):
        return lj.bsdf_queries(self.scene, q, variant)

    def light(self, q, variant=-1):
        return lj.light_queries(self.scene, q, variant)

    def sample_light(self, u):
        return lj.sample_light_queries(self.scene, u)

    def vertex(self, q, variant=-1):
        return lj.vertex_queries(self.scene, q, variant)

    def primary(self, q):
        return lj.primary_ray_queries(self.scene, q)

    def filter(self, q):
        return lj.filter_queries(self.ctx, q)

    def pcg32(self, streams, count, seed=0):
        return lj.pcg32_queries(self.ctx, streams, count, seed)

    def texture(self, q):
        return lj.texture_queries(self.scene, q)

    def frame(self, q):
        return lj.frame_queries(self.ctx, q)


def with_materials(hs, material_dicts):
    """Point a parsed scene's material table at a caller-built array (kept alive on the HostScene).  The shapes keep
    their material ids, so the table must be at least as long as the original."""
    import ctypes
    n0 = hs.desc.n_materials
    mats = [material_struct(m) for m in material_dicts]
    arr = (_abi.LjMaterial * max(len(mats), n0))()
    for i in range(n0):
        ctypes.memmove(ctypes.addressof(arr[i]), ctypes.addressof(hs.desc.materials[i]), ctypes.sizeof(_abi.LjMaterial))
    for i, m in enumerate(mats):
        ctypes.memmove(ctypes.addressof(arr[i]), ctypes.addressof(m), ctypes.sizeof(m))
    hs._material_override = arr
    hs.desc.materials = ctypes.cast(arr, ctypes.POINTER(_abi.LjMaterial))
    hs.desc.n_materials = len(arr)
    return hs
