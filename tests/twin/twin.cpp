// TEST INFRASTRUCTURE — host build of the *device* headers (device/dshade.h, device/dtrace.h) with g++.
//
// There is no GPU in the authoring container, so the float shading / traversal code that the HIP kernels run is
// also compiled for the CPU here and driven path by path, to debug it against the double-precision oracle before
// any GPU minute is spent.  This library is built only by the test suite, lives under tests/, is never loaded by
// the product and is not a fallback: lajolla_public_amd has no code path that reaches it.
#include "../../lajolla_public_amd/csrc/device/dshade.h"
#include "../../lajolla_public_amd/csrc/device/dvol.h"
#include "../../lajolla_public_amd/csrc/device/dtrace.h"
#include "../../lajolla_public_amd/csrc/host/flatten.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

using namespace ljd;

namespace {

struct HostMem {
    const DScene &sc;
    int stack[192];
    mutable unsigned long long n_nodes = 0, n_prims = 0;
    explicit HostMem(const DScene &s) : sc(s) {}
    DNode4 node(int i) const { n_nodes++; return sc.nodes[i]; }
    DPrim prim(int i) const { n_prims++; return sc.leaf_prims[i]; }
    const DSphere &sphere(int s) const { return sc.spheres[s]; }
    void push(int sp, int v) { stack[sp] = v; }
    int pop(int sp) const { return stack[sp]; }
    // the BVH8 twin of the same tree (dtrace.h traverse8)
    uint32_t stack8[2 * 64];
    const uint32_t *node8(int i) const { n_nodes++; return reinterpret_cast<const uint32_t *>(&sc.nodes8[i]); }
    void push8(int sp, uint32_t base, uint32_t bits) { stack8[2 * sp] = base; stack8[2 * sp + 1] = bits; }
    void pop8(int sp, uint32_t &base, uint32_t &bits) const { base = stack8[2 * sp]; bits = stack8[2 * sp + 1]; }
};

// traversal work counters (debugging aid): [shadow rays, shadow nodes, shadow prims, ext rays, ext nodes, ext prims]
static thread_local unsigned long long g_trav[8];
// (developer aid) number of leaf boxes of the whole tree a ray segment overlaps: the candidate count of a flat leaf scan
static int count_leaf_boxes(const DScene &sc, const RayF &ray) {
    const float ix = 1.0f / ray.dx, iy = 1.0f / ray.dy, iz = 1.0f / ray.dz;
    int n = 0;
    for (int i = 0; i < sc.n_nodes; i++) for (int k = 0; k < 4; k++) {
        if (sc.nodes[i].child[k] >= 0 || !(sc.nodes[i].lox[k] <= sc.nodes[i].hix[k])) continue;
        float te; if (box_test4(sc.nodes[i], k, ray, ix, iy, iz, ray.tfar, te)) n++;
    }
    return n;
}

// what k_extend does for one queue slot
void extend_one(const DScene &sc, PathState &ps) {
    HostMem mem(sc);
    RayF ray; ray.ox = ps.org.x; ray.oy = ps.org.y; ray.oz = ps.org.z;
    int code = 0;
    if (ps.stfar > 0.0f) {
        ray.dx = ps.sdir.x; ray.dy = ps.sdir.y; ray.dz = ps.sdir.z; ray.tnear = sc.eps; ray.tfar = ps.stfar;
        HitRec h;
        static const bool wide = getenv("LJ_TWIN_BVH8") != nullptr;   // developer aid: the same rays over the BVH8 (work counters below)
        if (!(wide ? traverse8<true>(mem, ray, h) : traverse<true>(mem, ray, h))) code |= HIT_VIS_BIT;
        if (getenv("LJ_TWIN_TRAV")) g_trav[6] += count_leaf_boxes(sc, ray);
        g_trav[0]++; g_trav[1] += mem.n_nodes; g_trav[2] += mem.n_prims; mem.n_nodes = mem.n_prims = 0;
    }
    float t = 0, u = 0, v = 0;
    if (!(ps.flags & PF_NO_EXT)) {
        ray.dx = ps.dir.x; ray.dy = ps.dir.y; ray.dz = ps.dir.z;
        ray.tnear = ((ps.flags & 0xffffu) == 2u) ? 0.0f : sc.eps; ray.tfar = INFINITY;
        HitRec h;
        static const bool wide = getenv("LJ_TWIN_BVH8") != nullptr;
        if (wide ? traverse8<false>(mem, ray, h) : traverse<false>(mem, ray, h)) { code |= (h.gprim + 1); t = h.t; u = h.u; v = h.v; }
        if (getenv("LJ_TWIN_TRAV")) g_trav[7] += count_leaf_boxes(sc, ray);
        g_trav[3]++; g_trav[4] += mem.n_nodes; g_trav[5] += mem.n_prims;
    }
    ps.ht = t; ps.hu = u; ps.hv = v; ps.hcode = code;
}

struct Twin { lj::FlatScene flat; DScene view; };

// what k_volpath's tracer does: one closest-hit query over the BVH
struct HostTracer {
    const DScene &sc;
    void tick(int) {}
    bool closest(f3 org, f3 dir, float tnear, float tfar, float &t, float &u, float &v, int &gprim) {
        HostMem mem(sc);
        RayF ray; ray.ox = org.x; ray.oy = org.y; ray.oz = org.z; ray.dx = dir.x; ray.dy = dir.y; ray.dz = dir.z; ray.tnear = tnear; ray.tfar = tfar;
        HitRec h;
        if (!traverse<false>(mem, ray, h)) return false;
        t = h.t; u = h.u; v = h.v; gprim = h.gprim;
        return true;
    }
};

} // namespace

extern "C" {

void *twin_create(const LjSceneDesc *d, char *err, int err_len) {
    try {
        Twin *t = new Twin();
        t->flat = lj::flatten_scene(*d);
        t->view = t->flat.host_view();
        return t;
    } catch (const std::exception &e) {
        if (err && err_len > 0) { strncpy(err, e.what(), err_len - 1); err[err_len - 1] = 0; }
        return nullptr;
    }
}
void twin_free(void *t) { delete (Twin *)t; }

void twin_tables(void *tv, double *bounds_radius, double *bounds_center, double *eps, double *light_pmf, double *light_cdf, double *light_power,
                 int *n_nodes, int *bvh_depth) {
    Twin *t = (Twin *)tv;
    *bounds_radius = t->flat.bounds_radius; for (int k = 0; k < 3; k++) bounds_center[k] = t->flat.bounds_center[k];
    *eps = t->flat.shadow_epsilon;
    for (size_t i = 0; i < t->flat.light_pmf_d.size(); i++) { light_pmf[i] = t->flat.light_pmf_d[i]; light_power[i] = t->flat.light_power_d[i]; }
    for (size_t i = 0; i < t->flat.light_cdf_d.size(); i++) light_cdf[i] = t->flat.light_cdf_d[i];
    *n_nodes = (int)t->flat.nodes.size(); *bvh_depth = t->flat.bvh_depth;
}

// per-sample radiance over a crop window, same layout as lj_render_samples
void twin_render_samples(void *tv, int spp, int max_depth, int use_max_depth, uint64_t seed, int x0, int y0, int x1, int y1, int n_threads, float *out,
                         unsigned long long *bounces_out) {
    Twin *t = (Twin *)tv;
    DScene sc = t->view;
    if (use_max_depth) sc.max_depth = max_depth;
    const int w = sc.cam.width, cw = x1 - x0, ch = y1 - y0;
    std::vector<uint32_t> pixels;
    for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) pixels.push_back((uint32_t)(y * w + x));
    DPass pass{}; pass.pixel_list = pixels.data(); pass.n_pixels = (uint32_t)pixels.size(); set_pass_divisors(pass, (uint32_t)spp, (uint32_t)sc.cam.width);
    pass.seed = seed ? seed : 0x853c49e6748fea9bULL; pass.sample_rgb = out;
    const uint64_t total = (uint64_t)cw * ch * spp;
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    std::vector<unsigned long long> bounces(n_threads, 0);
    auto worker = [&](int tid) {
        ShadeCounters cnt{};
        for (uint64_t s = tid; s < total; s += n_threads) {
            if (t->flat.integrator == LJ_INTEGRATOR_VOLPATH) {   // k_volpath: the whole path in one go
                HostTracer tr{sc};
                const uint32_t pixel = pixels[s / spp];
                uint32_t nb = 0;
                f3 r = vol_path_sample(sc, tr, (int)(pixel % (uint32_t)w), (int)(pixel / (uint32_t)w), (uint64_t)pixel * spp + (s % spp), pass.seed, nb);
                if (!(std::isfinite(r.x) && std::isfinite(r.y) && std::isfinite(r.z))) r = mk3(0, 0, 0);   // (volpath_body, kernels.hip)
                out[3 * s] = r.x; out[3 * s + 1] = r.y; out[3 * s + 2] = r.z;
                cnt.bounces += nb;
                continue;
            }
            PathState ps;
            generate_path(sc, pass, (uint32_t)s, ps);
            for (int step = 0; step < 100000; step++) {
                extend_one(sc, ps);
                if (!shade_path(sc, pass, ps, cnt)) break;
            }
            out[3 * s] = ps.rad.x; out[3 * s + 1] = ps.rad.y; out[3 * s + 2] = ps.rad.z;
        }
        bounces[tid] = cnt.bounces;
        if (getenv("LJ_TWIN_TRAV")) fprintf(stderr, "trav[%d]: shadow rays %llu nodes/ray %.2f prims/ray %.2f | ext rays %llu nodes/ray %.2f prims/ray %.2f | flat-scan candidates: %.2f per shadow ray, %.2f per ext ray\n", tid, g_trav[0], g_trav[1] / (double)g_trav[0], g_trav[2] / (double)g_trav[0], g_trav[3], g_trav[4] / (double)g_trav[3], g_trav[5] / (double)g_trav[3], g_trav[6] / (double)g_trav[0], g_trav[7] / (double)g_trav[3]);
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n_threads; i++) th.emplace_back(worker, i);
    worker(0);
    for (auto &x : th) x.join();
    unsigned long long b = 0; for (auto v : bounces) b += v;
    if (bounces_out) *bounces_out = b;
}

// the auxiliary "integrators" (k_aux): one value per pixel of the crop window, float[ch][cw][3]
void twin_aux(void *tv, int integrator, int x0, int y0, int x1, int y1, float *out) {
    Twin *t = (Twin *)tv;
    const DScene &sc = t->view;
    HostMem mem(sc);
    for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) {
        const f3 org = ld3(sc.cam.org), dir = camera_primary_dir(sc.cam, x, y, 0.5f, 0.5f);
        RayF ray; ray.ox = org.x; ray.oy = org.y; ray.oz = org.z; ray.dx = dir.x; ray.dy = dir.y; ray.dz = dir.z; ray.tnear = 0.0f; ray.tfar = INFINITY;
        HitRec h;
        traverse<false>(mem, ray, h);
        const f3 c = aux_value(sc, integrator, org, dir, h.t, h.u, h.v, h.gprim);
        float *o = out + 3 * ((size_t)(y - y0) * (x1 - x0) + (x - x0));
        o[0] = c.x; o[1] = c.y; o[2] = c.z;
    }
}

// ---- the per-object queries of include/lajolla_hip.h (lj_bsdf_queries ... lj_frame_queries), answered by the host build of
// the same device functions queries.hip calls: lets the CPU suite run the very comparisons the GPU suite makes
// (tests/test_device_kats.py), so that a failure on the GPU box is the device's arithmetic, not the test's logic.
static DVertex vertex_in(const LjVertex &v) {
    DVertex vx;
    vx.position = ld3(v.position); vx.gn = ld3(v.geometry_normal);
    vx.frame.x = ld3(v.frame_x); vx.frame.y = ld3(v.frame_y); vx.frame.n = ld3(v.frame_n);
    vx.u = v.uv[0]; vx.v = v.uv[1]; vx.uv_screen_size = v.uv_screen_size;
    vx.material_id = v.material_id; vx.light_id = v.light_id; vx.gprim = 0; vx.is_sphere = false;
    return vx;
}
static void st3(float *o, f3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

void twin_bsdf_queries(void *tv, int variant, int64_t n, const LjBsdfQuery *q, LjBsdfResult *r) {
    const DScene &sc = ((Twin *)tv)->view;
    with_shade_variant(variant, [&](auto ft) {
        using Ft = decltype(ft);
        for (int64_t i = 0; i < n; i++) {
            const DVertex vx = vertex_in(q[i].vertex);
            const DMaterial &m = sc.materials[vx.material_id];
            const f3 din = ld3(q[i].dir_in), dout = ld3(q[i].dir_out);
            f3 f; float pdf;
            bsdf_eval_pdf<Ft>(sc, m, din, dout, vx, f, pdf);
            const BsdfSample bs = bsdf_sample<Ft>(sc, m, din, vx, q[i].rnd_uv[0], q[i].rnd_uv[1], q[i].rnd_w);
            st3(r[i].eval, f); r[i].pdf = pdf; st3(r[i].sample_dir, bs.dir_out); r[i].sample_eta = bs.eta; r[i].sample_roughness = bs.roughness; r[i].sample_valid = bs.valid ? 1 : 0;
        }
    });
}
void twin_light_queries(void *tv, int variant, int64_t n, const LjLightQuery *q, LjLightResult *r) {
    const DScene &sc = ((Twin *)tv)->view;
    with_shade_variant(variant, [&](auto ft) {
        using Ft = decltype(ft);
        for (int64_t i = 0; i < n; i++) {
            const DLight &L = sc.lights[q[i].light_id];
            const f3 ref = ld3(q[i].ref);
            const LightSample ls = sample_point_on_light<Ft>(sc, L, ref, q[i].rnd_uv[0], q[i].rnd_uv[1], q[i].rnd_w);
            st3(r[i].position, ls.position); st3(r[i].normal, ls.normal);
            r[i].pdf = pdf_point_on_light<Ft>(sc, L, ls.position, ls.normal, ref);
            st3(r[i].emission, light_emission<Ft>(sc, L, ld3(q[i].view_dir), ls.normal));
            r[i].pmf = L.pmf; r[i]._pad = 0;
            for (int k = 0; k < 3; k++) r[i].position_d[k] = ls.dpos[k];
        }
    });
}
void twin_sample_light_queries(void *tv, int64_t n, const float *u, int32_t *id) {
    const DScene &sc = ((Twin *)tv)->view;
    for (int64_t i = 0; i < n; i++) id[i] = sample_cdf(sc.light_cdf, sc.n_lights, u[i]);
}
void twin_vertex_queries(void *tv, int variant, int64_t n, const LjHitQuery *q, LjHitResult *r) {
    Twin *t = (Twin *)tv;
    const DScene &sc = t->view;
    // global primitive id of a (shape, primitive): the shape's first global id + primitive
    std::vector<int64_t> first;
    {   int64_t best = -1; (void)best;
        int n_shapes = 0; for (const auto &p : t->flat.prims) n_shapes = p.shape_id + 1 > n_shapes ? p.shape_id + 1 : n_shapes;
        first.assign(n_shapes + 1, -1);
        for (size_t g = 0; g < t->flat.prims.size(); g++) if (first[t->flat.prims[g].shape_id] < 0) first[t->flat.prims[g].shape_id] = (int64_t)g;
    }
    with_shade_variant(variant, [&](auto ft) {
        using Ft = decltype(ft);
        for (int64_t i = 0; i < n; i++) {
            const int gprim = (int)(first[q[i].shape_id] + q[i].primitive_id);
            const f3 org = ld3(q[i].org), dir = ld3(q[i].dir);
            const DVertex vx = build_vertex(sc, org, dir, q[i].t, q[i].u, q[i].v, gprim, q[i].ray_spread);
            LjHitResult &o = r[i];
            st3(o.vertex.position, vx.position); st3(o.vertex.geometry_normal, vx.gn);
            st3(o.vertex.frame_x, vx.frame.x); st3(o.vertex.frame_y, vx.frame.y); st3(o.vertex.frame_n, vx.frame.n);
            o.vertex.uv_screen_size = vx.uv_screen_size;
            DScene s1 = sc; s1.init_spread = q[i].ray_spread;
            o.vertex.mean_curvature = aux_value(s1, 2, org, dir, q[i].t, q[i].u, q[i].v, gprim).x;
            o.vertex.uv[0] = vx.u; o.vertex.uv[1] = vx.v;
            o.vertex.material_id = vx.material_id; o.vertex.light_id = vx.light_id;
            o.vertex.shape_id = sc.prims[gprim].shape_id; o.vertex.primitive_id = sc.prims[gprim].prim_id;
            f3 em = mk3(0, 0, 0);
            if (vx.light_id >= 0) em = light_emission<Ft>(sc, sc.lights[vx.light_id], -dir, vx.gn);
            st3(o.emission, em); o._pad = 0;
        }
    });
}
void twin_primary_ray_queries(void *tv, int64_t n, const LjPrimaryQuery *q, LjPrimaryResult *r) {
    const DScene &sc = ((Twin *)tv)->view;
    for (int64_t i = 0; i < n; i++) { st3(r[i].org, ld3(sc.cam.org)); st3(r[i].dir, camera_primary_dir(sc.cam, q[i].x, q[i].y, q[i].jx, q[i].jy)); }
}
void twin_filter_queries(int64_t n, const LjFilterQuery *q, float *out) {
    for (int64_t i = 0; i < n; i++) filter_sample(q[i].kind, q[i].param, q[i].rnd[0], q[i].rnd[1], out[2 * i], out[2 * i + 1]);
}
void twin_pcg32_queries(int64_t n, const uint64_t *streams, uint64_t seed, int count, uint32_t *u32, float *real) {
    for (int64_t i = 0; i < n; i++) {
        const uint64_t inc = pcg32_inc(streams[i]);
        uint64_t st = pcg32_init(streams[i], seed ? seed : 0x853c49e6748fea9bULL), st2 = st;
        for (int k = 0; k < count; k++) { u32[i * count + k] = pcg32_next(st, inc); if (real) real[i * count + k] = pcg32_real(st2, inc); }
    }
}
void twin_texture_queries(void *tv, int64_t n, const LjTextureQuery *q, float *rgb) {
    const DScene &sc = ((Twin *)tv)->view;
    for (int64_t i = 0; i < n; i++) {
        const LjTexture &t = q[i].texture;
        DTexture d{}; d.kind = t.kind; d.texture_id = t.texture_id;
        for (int k = 0; k < 3; k++) { d.value[k] = (float)t.value[k]; d.color1[k] = (float)t.color1[k]; }
        d.uscale = (float)t.uscale; d.vscale = (float)t.vscale; d.uoffset = (float)t.uoffset; d.voffset = (float)t.voffset;
        st3(rgb + 3 * i, eval_texture<FeatAll>(sc, d, q[i].spectrum != 0, q[i].uv[0], q[i].uv[1], q[i].footprint));
    }
}
void twin_frame_queries(int64_t n, const LjFrameQuery *q, LjFrameResult *r) {
    for (int64_t i = 0; i < n; i++) {
        const Frame3 f = make_frame(ld3(q[i].n));
        const f3 v = ld3(q[i].v);
        st3(r[i].x, f.x); st3(r[i].y, f.y); st3(r[i].to_local, to_local(f, v)); st3(r[i].to_world, to_world(f, v));
    }
}
int twin_shade_variant(void *tv) {
    Twin *t = (Twin *)tv;
    uint32_t kinds = 0; bool textured = false, sphere_lights = false;
    for (const auto &m : t->flat.materials) { kinds |= 1u << m.kind; for (int k = 0; k < 12; k++) textured = textured || m.tex[k].kind != 0; }
    for (const auto &l : t->flat.lights) sphere_lights = sphere_lights || (l.kind == 0 && l.is_sphere);
    for (int v = 0; v < kNumShadeVariants; v++) if (variant_covers(v, kinds, textured, t->flat.envmap_light_id >= 0, sphere_lights)) return v;
    return kNumShadeVariants - 1;
}
int twin_variant_covers(void *tv, int v) {
    Twin *t = (Twin *)tv;
    uint32_t kinds = 0; bool textured = false, sphere_lights = false;
    for (const auto &m : t->flat.materials) { kinds |= 1u << m.kind; for (int k = 0; k < 12; k++) textured = textured || m.tex[k].kind != 0; }
    for (const auto &l : t->flat.lights) sphere_lights = sphere_lights || (l.kind == 0 && l.is_sphere);
    return variant_covers(v, kinds, textured, t->flat.envmap_light_id >= 0, sphere_lights) ? 1 : 0;
}

// BVH shape (developer aid): out[0] nodes, out[1] leaves, out[2] depth, out[3..10] leaves holding 1..8 primitives
void twin_bvh_stats(void *tv, long long *out) {
    Twin *t = (Twin *)tv;
    for (int i = 0; i < 11; i++) out[i] = 0;
    out[0] = (long long)t->flat.nodes.size(); out[2] = t->flat.bvh_depth;
    for (const auto &nd : t->flat.nodes) for (int k = 0; k < 4; k++) {
        if (!(nd.lox[k] <= nd.hix[k])) continue;   // empty slot
        if (nd.child[k] < 0) { out[1]++; out[3 + ((~nd.child[k]) & 7)]++; }
    }
}

void twin_intersect(void *tv, int64_t n, const LjRay *rays, LjHit *hits) {
    Twin *t = (Twin *)tv;
    const DScene &sc = t->view;
    HostMem mem(sc);
    for (int64_t i = 0; i < n; i++) {
        RayF r; r.ox = rays[i].org[0]; r.oy = rays[i].org[1]; r.oz = rays[i].org[2]; r.dx = rays[i].dir[0]; r.dy = rays[i].dir[1]; r.dz = rays[i].dir[2];
        r.tnear = rays[i].tnear; r.tfar = rays[i].tfar;
        HitRec h; LjHit o{0, 0, 0, -1, -1};
        if (traverse<false>(mem, r, h)) { const DPrimShade &ps = sc.prims[h.gprim]; o = LjHit{h.t, h.u, h.v, ps.shape_id, ps.prim_id}; }
        hits[i] = o;
    }
}
// the same queries over the BVH8 (DNode8) of the scene; work[0..1] += node steps, primitive tests
void twin_intersect8(void *tv, int64_t n, const LjRay *rays, LjHit *hits, unsigned long long *work) {
    Twin *t = (Twin *)tv;
    const DScene &sc = t->view;
    HostMem mem(sc);
    for (int64_t i = 0; i < n; i++) {
        RayF r; r.ox = rays[i].org[0]; r.oy = rays[i].org[1]; r.oz = rays[i].org[2]; r.dx = rays[i].dir[0]; r.dy = rays[i].dir[1]; r.dz = rays[i].dir[2];
        r.tnear = rays[i].tnear; r.tfar = rays[i].tfar;
        HitRec h; LjHit o{0, 0, 0, -1, -1};
        if (traverse8<false>(mem, r, h)) { const DPrimShade &ps = sc.prims[h.gprim]; o = LjHit{h.t, h.u, h.v, ps.shape_id, ps.prim_id}; }
        hits[i] = o;
    }
    if (work) { work[0] += mem.n_nodes; work[1] += mem.n_prims; }
}
void twin_occluded8(void *tv, int64_t n, const LjRay *rays, uint8_t *occ) {
    Twin *t = (Twin *)tv;
    HostMem mem(t->view);
    for (int64_t i = 0; i < n; i++) {
        RayF r; r.ox = rays[i].org[0]; r.oy = rays[i].org[1]; r.oz = rays[i].org[2]; r.dx = rays[i].dir[0]; r.dy = rays[i].dir[1]; r.dz = rays[i].dir[2];
        r.tnear = rays[i].tnear; r.tfar = rays[i].tfar;
        HitRec h; occ[i] = traverse8<true>(mem, r, h) ? 1 : 0;
    }
}
void twin_intersect_work(void *tv, int64_t n, const LjRay *rays, unsigned long long *work) {   // BVH4 node steps / primitive tests of the same rays
    Twin *t = (Twin *)tv;
    HostMem mem(t->view);
    for (int64_t i = 0; i < n; i++) {
        RayF r; r.ox = rays[i].org[0]; r.oy = rays[i].org[1]; r.oz = rays[i].org[2]; r.dx = rays[i].dir[0]; r.dy = rays[i].dir[1]; r.dz = rays[i].dir[2];
        r.tnear = rays[i].tnear; r.tfar = rays[i].tfar;
        HitRec h; traverse<false>(mem, r, h);
    }
    work[0] += mem.n_nodes; work[1] += mem.n_prims;
}
// fast_div (dmath.h) against the machine's division: every divisor of `divs`, numerators at the ends of the range, around every multiple
// of the divisor near them, and random ones.  Returns the number of disagreements.
long long twin_fast_div_mismatches(const uint32_t *divs, int n_divs, int n_random) {
    long long bad = 0;
    uint64_t st = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < n_divs; i++) {
        const uint32_t d = divs[i];
        const DFastDiv f = make_fast_div(d);
        auto check = [&](uint32_t n) { if (fast_div(n, f) != n / d) bad++; };
        for (uint32_t k = 0; k < 4; k++) { check(k); check(0xffffffffu - k); check(0x80000000u - k); check(0x80000000u + k); }
        for (uint64_t q : {1ull, 2ull, 3ull, (0xffffffffull / d) / 2, 0xffffffffull / d - 1, 0xffffffffull / d}) {
            const uint64_t m = q * d;
            for (long long o = -2; o <= 2; o++) { const long long n = (long long)m + o; if (n >= 0 && n <= 0xffffffffll) check((uint32_t)n); }
        }
        for (int r = 0; r < n_random; r++) { st = st * 6364136223846793005ULL + 1442695040888963407ULL; check((uint32_t)(st >> 32)); check((uint32_t)(st >> 45)); }
    }
    return bad;
}

// sample_cdf_guided against the full bisection on every environment-map table of the scene: random u, and the values around every
// bin edge (b / n and its float neighbours) where a guide that is one entry short would show.  Returns the number of disagreements.
long long twin_cdf_guide_mismatches(void *tv, int n_random) {
    const DScene &sc = ((Twin *)tv)->view;
    long long bad = 0, checked = 0;
    uint64_t st = 0x1234567ull;
    auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (float)((st >> 40) * (1.0 / 16777216.0)); };
    auto check = [&](const float *cdf, const float *guide, int n, float u) {
        if (!(u >= 0.0f && u < 1.0f)) return;
        checked++;
        const int want = sample_cdf(cdf, n, u);
        if (want != sample_cdf_guided(cdf, guide, n, u)) bad++;
        float c0 = -1.0f, c1 = -1.0f;
        if (want != sample_cdf_guided(cdf, guide, n, u, c0, c1) || c0 != cdf[want] || c1 != cdf[want + 1]) bad++;
    };
    auto table = [&](const float *cdf, const float *guide, int n) {
        for (int i = 0; i < n_random; i++) check(cdf, guide, n, rnd());
        for (int b = 0; b <= n; b++) { const float e = (float)b / (float)n; check(cdf, guide, n, e); check(cdf, guide, n, nextafterf(e, 0.0f)); check(cdf, guide, n, nextafterf(e, 2.0f)); }
        for (int i = 0; i <= n; i++) { check(cdf, guide, n, cdf[i]); check(cdf, guide, n, nextafterf(cdf[i], 0.0f)); check(cdf, guide, n, nextafterf(cdf[i], 2.0f)); }
    };
    for (int li = 0; li < sc.n_lights; li++) {
        const DLight &L = sc.lights[li];
        if (L.kind == 0) continue;
        table(sc.env_tables + L.env_cdf_marg, sc.env_tables + L.env_guide_marg, L.env_h);
        for (int y = 0; y < L.env_h; y += (L.env_h > 64 ? L.env_h / 64 : 1))
            table(sc.env_tables + L.env_cdf_rows + (int64_t)y * (L.env_w + 1), sc.env_tables + L.env_guide_rows + (int64_t)y * L.env_w, L.env_w);
    }
    return checked > 0 ? bad : -1;
}
void twin_bvh8_info(void *tv, long long *out) {   // nodes, depth, filled slots, leaf slots
    Twin *t = (Twin *)tv;
    out[0] = (long long)t->flat.nodes8.size(); out[1] = t->flat.bvh8_depth; out[2] = out[3] = 0;
    for (const auto &nd : t->flat.nodes8) for (int s = 0; s < 8; s++) {
        if (nd.imask & (1u << s)) out[2]++;
        else if (nd.meta[s] & 0x80u) { out[2]++; out[3]++; }
    }
}
void twin_occluded(void *tv, int64_t n, const LjRay *rays, uint8_t *occ) {
    Twin *t = (Twin *)tv;
    HostMem mem(t->view);
    for (int64_t i = 0; i < n; i++) {
        RayF r; r.ox = rays[i].org[0]; r.oy = rays[i].org[1]; r.oz = rays[i].org[2]; r.dx = rays[i].dir[0]; r.dy = rays[i].dir[1]; r.dz = rays[i].dir[2];
        r.tnear = rays[i].tnear; r.tfar = rays[i].tfar;
        HitRec h; occ[i] = traverse<true>(mem, r, h) ? 1 : 0;
    }
}

} // extern "C"
