// C-ABI entry points that own the GPU: context, scene upload, the wavefront render loop, batched ray queries.
// One context = one HIP device + one stream; one process per GPU (the multi-GPU layer above this is
// torch.distributed over RCCL, see bench.py).  Every HIP call is checked; failures surface as LJ_ERR_DEVICE with
// the HIP error string — there is no CPU fallback behind any of these functions.
#include <hip/hip_runtime.h>
#include "../host/api_common.h"
#include "../host/flatten.h"
#include "dtypes.h"
#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>

namespace ljd {
bool scene_is_small(int n_nodes, int n_prims, int bvh_depth);
int large_stack_depth();
void launch_prepare(DCtrl *c, hipStream_t s);
void launch_generate(const DScene &sc, const DPass &pass, const DQueue &q, const DCtrl *c, int grid, hipStream_t s);
void launch_extend(const DScene &sc, const DQueue &q, DCtrl *c, bool small, int grid, hipStream_t s);
void launch_shade(const DScene &sc, const DPass &pass, const DQueue &qin, const DQueue &qout, DCtrl *c, int grid, hipStream_t s);
void launch_resolve(const DPass &pass, uint32_t n_pixels, float *rgb, hipStream_t s);
void launch_trace_rays(const DScene &sc, const void *rays, long long n, void *hits, unsigned char *occ, bool small, int grid, hipStream_t s);
}

namespace {

using lj::LjError;

#define HIP_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) \
    throw LjError(LJ_ERR_DEVICE, std::string(#expr) + " failed: " + hipGetErrorString(_e)); } while (0)

struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    void alloc(size_t n) { release(); if (n) { HIP_CHECK(hipMalloc(&p, n)); bytes = n; } }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    ~DevBuf() { release(); }
    DevBuf() = default; DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
};

template <typename T> void upload(DevBuf &b, const std::vector<T> &v, hipStream_t s) {
    b.alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (!v.empty()) HIP_CHECK(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

constexpr size_t kQueueSlotBytes = 124;
constexpr int kQueueArrays = 29;

} // namespace

struct lj_context {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cus = 256;
    // workspace, grown on demand and reused across renders
    DevBuf queue_mem[2]; uint32_t queue_capacity = 0;
    DevBuf ctrl, sample_rgb, pixel_list, frame;
    ljd::DCtrl *ctrl_host = nullptr;  // pinned
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
};

struct lj_scene {
    lj_context *ctx = nullptr;
    lj::FlatScene flat;  // host copy (tables for lj_scene_info; arrays already uploaded)
    DevBuf nodes, leaf_prims, prims, spheres, materials, lights, light_cdf, light_tris, light_tri_cdf, images3, images1, texels, env_tables;
    ljd::DScene dscene{};
    bool small = false;
    LjStats stats{};
};

namespace {

void set_device(lj_context *ctx) { HIP_CHECK(hipSetDevice(ctx->device)); }

ljd::DQueue carve_queue(void *base, uint32_t cap) {
    // one allocation, 29 arrays; every array starts on a 256-byte boundary so wave accesses are aligned
    ljd::DQueue q{};
    char *p = (char *)base;
    auto take = [&](size_t elem) { void *r = p; size_t bytes = ((size_t)cap * elem + 255) & ~(size_t)255; p += bytes; return r; };
    q.ox = (float *)take(4); q.oy = (float *)take(4); q.oz = (float *)take(4);
    q.dx = (float *)take(4); q.dy = (float *)take(4); q.dz = (float *)take(4);
    q.ht = (float *)take(4); q.hu = (float *)take(4); q.hv = (float *)take(4); q.hprim = (int32_t *)take(4);
    q.sx = (float *)take(4); q.sy = (float *)take(4); q.sz = (float *)take(4); q.st = (float *)take(4);
    q.wr = (float *)take(4); q.wg = (float *)take(4); q.wb = (float *)take(4); q.rr = (float *)take(4); q.p2 = (float *)take(4);
    q.lr = (float *)take(4); q.lg = (float *)take(4); q.lb = (float *)take(4);
    q.nr = (float *)take(4); q.ng = (float *)take(4); q.nb = (float *)take(4);
    q.sample = (uint32_t *)take(4); q.rng = (uint64_t *)take(8);
    q.eta_scale = (float *)take(4); q.spread = (float *)take(4); q.flags = (uint32_t *)take(4);
    return q;
}
size_t queue_bytes(uint32_t cap) { return (size_t)31 * ((((size_t)cap * 4 + 255) & ~(size_t)255)) + 256; }  // 30 arrays, rng counts twice

void ensure_queues(lj_context *ctx, uint32_t cap) {
    if (ctx->queue_capacity >= cap) return;
    for (int i = 0; i < 2; i++) ctx->queue_mem[i].alloc(queue_bytes(cap));
    ctx->queue_capacity = cap;
}

struct RenderPlan {
    int spp; uint32_t pool; uint64_t seed;
    std::vector<uint32_t> pixels;  // rendered pixels in tile order
    int max_depth;
};

RenderPlan make_plan(const lj_scene *sc, const LjRenderArgs *a) {
    RenderPlan p;
    const int w = sc->flat.cam.width, h = sc->flat.cam.height;
    p.spp = (a && a->spp > 0) ? a->spp : sc->flat.spp;
    if (p.spp <= 0) throw LjError(LJ_ERR_INVALID_ARG, "samples per pixel must be positive");
    p.seed = (a && a->seed) ? a->seed : 0x853c49e6748fea9bULL;
    p.max_depth = (a && a->max_depth != INT32_MIN) ? a->max_depth : sc->flat.max_depth;
    p.pool = (a && a->pool_paths) ? a->pool_paths : (1u << 21);
    p.pool = std::max<uint32_t>(p.pool, 4096);
    if (a && a->rng_mode != LJ_RNG_SAMPLE) throw LjError(LJ_ERR_UNSUPPORTED, "only LJ_RNG_SAMPLE exists on the device (a per-tile sequential stream cannot be parallelised, SURVEY §0.2)");
    int rank = a ? a->rank : 0, world = (a && a->world_size > 0) ? a->world_size : 1;
    if (rank < 0 || rank >= world) throw LjError(LJ_ERR_INVALID_ARG, "rank must be in [0, world_size)");
    bool crop = a && a->crop_x1 > a->crop_x0 && a->crop_y1 > a->crop_y0;
    int cx0 = crop ? a->crop_x0 : 0, cy0 = crop ? a->crop_y0 : 0, cx1 = crop ? a->crop_x1 : w, cy1 = crop ? a->crop_y1 : h;
    if (cx0 < 0 || cy0 < 0 || cx1 > w || cy1 > h) throw LjError(LJ_ERR_INVALID_ARG, "crop window outside the film");
    const int tile = 16, ntx = (w + tile - 1) / tile, nty = (h + tile - 1) / tile;  // render.cpp:75-77
    if (crop) {  // crop windows are enumerated row-major (lj_render_samples layout)
        for (int y = cy0; y < cy1; y++) for (int x = cx0; x < cx1; x++) {
            int t = (y / tile) * ntx + (x / tile);
            if (t % world == rank) p.pixels.push_back((uint32_t)(y * w + x));
        }
    } else {
        for (int t = 0; t < ntx * nty; t++) {
            if (t % world != rank) continue;
            int tx = t % ntx, ty = t / ntx;
            int x0 = tx * tile, x1 = std::min(x0 + tile, w), y0 = ty * tile, y1 = std::min(y0 + tile, h);
            for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) p.pixels.push_back((uint32_t)(y * w + x));
        }
    }
    return p;
}

// Renders plan.pixels; if rgb_dev != null writes radiance/spp there (other pixels untouched), if samples_dev_out != null
// the per-sample radiance of every pass is copied to host memory `samples_host` in pixel-list order.
void run_render(lj_scene *sc, const RenderPlan &plan, float *rgb_dev, float *samples_host, hipStream_t stream, bool timing) {
    lj_context *ctx = sc->ctx;
    set_device(ctx);
    ljd::DScene ds = sc->dscene;
    ds.max_depth = plan.max_depth;
    const uint64_t n_pix = plan.pixels.size();
    LjStats &st = sc->stats; st = LjStats{};
    if (n_pix == 0) return;
    // pass size: keep the per-sample radiance buffer <= ~1.5 GiB and sample ids in 32 bits
    const uint64_t max_samples_pass = (uint64_t)1 << 27;
    uint64_t pix_per_pass = std::max<uint64_t>(1, max_samples_pass / (uint64_t)plan.spp);
    pix_per_pass = std::min<uint64_t>(pix_per_pass, n_pix);
    const uint32_t cap = (uint32_t)std::min<uint64_t>(plan.pool, pix_per_pass * (uint64_t)plan.spp);
    ensure_queues(ctx, cap);
    ljd::DQueue q[2] = {carve_queue(ctx->queue_mem[0].p, ctx->queue_capacity), carve_queue(ctx->queue_mem[1].p, ctx->queue_capacity)};
    if (ctx->sample_rgb.bytes < pix_per_pass * plan.spp * 12) ctx->sample_rgb.alloc(pix_per_pass * plan.spp * 12);
    if (ctx->pixel_list.bytes < n_pix * 4) ctx->pixel_list.alloc(n_pix * 4);
    HIP_CHECK(hipMemcpyAsync(ctx->pixel_list.p, plan.pixels.data(), n_pix * 4, hipMemcpyHostToDevice, stream));
    ljd::DCtrl *dctrl = (ljd::DCtrl *)ctx->ctrl.p;
    const int grid_stream = ctx->n_cus * 4, grid_extend = ctx->n_cus * (sc->small ? 4 : 2);
    HIP_CHECK(hipEventRecord(ctx->ev_begin, stream));
    double extend_ms = 0, shade_ms = 0;
    for (uint64_t p0 = 0; p0 < n_pix; p0 += pix_per_pass) {
        const uint64_t np = std::min<uint64_t>(pix_per_pass, n_pix - p0);
        ljd::DPass pass{};
        pass.pixel_list = (const uint32_t *)ctx->pixel_list.p + p0; pass.n_pixels = (uint32_t)np; pass.spp = (uint32_t)plan.spp;
        pass.seed = plan.seed; pass.sample_rgb = (float *)ctx->sample_rgb.p;
        ljd::DCtrl init{}; init.total_samples = np * (uint64_t)plan.spp; init.capacity = cap;
        init.bounce_iterations = st.bounce_iterations; init.rays_closest = st.rays_closest; init.rays_shadow = st.rays_shadow;
        init.samples_done = 0; init.steps = (uint32_t)st.wavefront_steps; init.path_steps = 0;
        *ctx->ctrl_host = init;
        HIP_CHECK(hipMemcpyAsync(dctrl, ctx->ctrl_host, sizeof(ljd::DCtrl), hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        int cur = 0;
        uint64_t path_steps_total = 0;
        for (int guard = 0; guard < (1 << 22); guard++) {
            // a batch of steps runs without any host round trip; the device-side control block carries all counts
            const int batch = 8;
            for (int b = 0; b < batch; b++) {
                ljd::launch_prepare(dctrl, stream);
                ljd::launch_generate(ds, pass, q[cur], dctrl, grid_stream, stream);
                if (timing) HIP_CHECK(hipEventRecord(ctx->ev_k0, stream));
                ljd::launch_extend(ds, q[cur], dctrl, sc->small, grid_extend, stream);
                if (timing) { HIP_CHECK(hipEventRecord(ctx->ev_k1, stream)); }
                ljd::launch_shade(ds, pass, q[cur], q[cur ^ 1], dctrl, grid_stream, stream);
                if (timing) {
                    HIP_CHECK(hipEventRecord(ctx->ev_end, stream));
                    HIP_CHECK(hipEventSynchronize(ctx->ev_end));
                    float a = 0, c = 0;
                    HIP_CHECK(hipEventElapsedTime(&a, ctx->ev_k0, ctx->ev_k1)); HIP_CHECK(hipEventElapsedTime(&c, ctx->ev_k1, ctx->ev_end));
                    extend_ms += a; shade_ms += c; st.extend_launches++; st.shade_launches++;
                }
                cur ^= 1;
            }
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpyAsync(ctx->ctrl_host, dctrl, sizeof(ljd::DCtrl), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            const ljd::DCtrl &c = *ctx->ctrl_host;
            path_steps_total = c.path_steps;
            if (c.next_sample >= c.total_samples && c.n_out == 0) break;
        }
        const ljd::DCtrl &c = *ctx->ctrl_host;
        if (c.samples_done != c.total_samples)
            throw LjError(LJ_ERR_INTERNAL, "wavefront loop ended with " + std::to_string(c.samples_done) + " of " + std::to_string(c.total_samples) + " samples finished");
        st.samples += c.total_samples; st.bounce_iterations = c.bounce_iterations; st.rays_closest = c.rays_closest; st.rays_shadow = c.rays_shadow;
        st.wavefront_steps = c.steps;
        // algorithmic queue traffic: extend reads 48 + writes 16 per path-step, shade reads 108 and writes 108 per survivor
        st.extend_bytes += path_steps_total * 64ull;
        st.shade_bytes += path_steps_total * 108ull + (path_steps_total - c.total_samples) * 108ull + c.total_samples * 12ull;
        if (rgb_dev) ljd::launch_resolve(pass, (uint32_t)np, rgb_dev, stream);
        if (samples_host) {
            HIP_CHECK(hipMemcpyAsync(samples_host + p0 * (uint64_t)plan.spp * 3, ctx->sample_rgb.p, np * (uint64_t)plan.spp * 12, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
        }
    }
    HIP_CHECK(hipEventRecord(ctx->ev_end, stream));
    HIP_CHECK(hipEventSynchronize(ctx->ev_end));
    float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
    st.render_ms = ms; st.extend_ms = extend_ms; st.shade_ms = shade_ms;
    st.queue_bytes = st.extend_bytes + st.shade_bytes;
}

} // namespace

extern "C" {

int lj_context_create(int device_id, lj_context **out) {
    return lj::guard([&]() {
        if (!out) throw LjError(LJ_ERR_INVALID_ARG, "lj_context_create: null out");
        *out = nullptr;
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n == 0) throw LjError(LJ_ERR_DEVICE, std::string("no HIP device available (") + hipGetErrorString(e) + "); this library has no CPU path");
        if (device_id < 0 || device_id >= n) throw LjError(LJ_ERR_INVALID_ARG, "device id out of range");
        auto ctx = std::make_unique<lj_context>();
        ctx->device = device_id;
        HIP_CHECK(hipSetDevice(device_id));
        hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            throw LjError(LJ_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + "; this build contains gfx950 code objects only");
        ctx->n_cus = prop.multiProcessorCount;
        HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->ctrl.alloc(sizeof(ljd::DCtrl));
        HIP_CHECK(hipHostMalloc((void **)&ctx->ctrl_host, sizeof(ljd::DCtrl), hipHostMallocDefault));
        HIP_CHECK(hipEventCreate(&ctx->ev_begin)); HIP_CHECK(hipEventCreate(&ctx->ev_end));
        HIP_CHECK(hipEventCreate(&ctx->ev_k0)); HIP_CHECK(hipEventCreate(&ctx->ev_k1));
        *out = ctx.release();
    });
}

void lj_context_destroy(lj_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    if (ctx->ctrl_host) (void)hipHostFree(ctx->ctrl_host);
    for (hipEvent_t e : {ctx->ev_begin, ctx->ev_end, ctx->ev_k0, ctx->ev_k1}) if (e) (void)hipEventDestroy(e);
    delete ctx;
}

int lj_scene_upload(lj_context *ctx, const LjSceneDesc *desc, lj_scene **out) {
    return lj::guard([&]() {
        if (!ctx || !desc || !out) throw LjError(LJ_ERR_INVALID_ARG, "lj_scene_upload: null argument");
        *out = nullptr;
        set_device(ctx);
        auto sc = std::make_unique<lj_scene>();
        sc->ctx = ctx;
        sc->flat = lj::flatten_scene(*desc);
        lj::FlatScene &F = sc->flat;
        hipStream_t s = ctx->stream;
        upload(sc->nodes, F.nodes, s); upload(sc->leaf_prims, F.leaf_prims, s); upload(sc->prims, F.prims, s); upload(sc->spheres, F.spheres, s);
        upload(sc->materials, F.materials, s); upload(sc->lights, F.lights, s); upload(sc->light_cdf, F.light_cdf, s);
        upload(sc->light_tris, F.light_tris, s); upload(sc->light_tri_cdf, F.light_tri_cdf, s);
        upload(sc->images3, F.images3, s); upload(sc->images1, F.images1, s); upload(sc->texels, F.texels, s); upload(sc->env_tables, F.env_tables, s);
        HIP_CHECK(hipStreamSynchronize(s));
        ljd::DScene d = F.host_view();
        d.nodes = (const ljd::DNode *)sc->nodes.p; d.leaf_prims = (const ljd::DPrim *)sc->leaf_prims.p; d.prims = (const ljd::DPrimShade *)sc->prims.p;
        d.spheres = (const ljd::DSphere *)sc->spheres.p; d.materials = (const ljd::DMaterial *)sc->materials.p; d.lights = (const ljd::DLight *)sc->lights.p;
        d.light_cdf = (const float *)sc->light_cdf.p; d.light_tris = (const ljd::DLightTri *)sc->light_tris.p; d.light_tri_cdf = (const float *)sc->light_tri_cdf.p;
        d.images3 = (const ljd::DImage *)sc->images3.p; d.images1 = (const ljd::DImage *)sc->images1.p; d.texels = (const float *)sc->texels.p; d.env_tables = (const float *)sc->env_tables.p;
        sc->dscene = d;
        sc->small = ljd::scene_is_small((int)F.nodes.size(), (int)F.leaf_prims.size(), F.bvh_depth);
        if (!sc->small && F.bvh_depth > ljd::large_stack_depth())
            throw LjError(LJ_ERR_INTERNAL, "BVH depth " + std::to_string(F.bvh_depth) + " exceeds the traversal stack");
        *out = sc.release();
    });
}

void lj_scene_destroy(lj_scene *scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->ctx->device);
    (void)hipStreamSynchronize(scene->ctx->stream);
    delete scene;
}

int lj_render_device(lj_scene *scene, const LjRenderArgs *args, float *rgb_device, void *hip_stream) {
    return lj::guard([&]() {
        if (!scene || !rgb_device) throw LjError(LJ_ERR_INVALID_ARG, "lj_render_device: null argument");
        set_device(scene->ctx);
        hipStream_t s = hip_stream ? (hipStream_t)hip_stream : scene->ctx->stream;
        RenderPlan plan = make_plan(scene, args);
        const size_t fb = (size_t)scene->flat.cam.width * scene->flat.cam.height * 3 * sizeof(float);
        HIP_CHECK(hipMemsetAsync(rgb_device, 0, fb, s));
        run_render(scene, plan, rgb_device, nullptr, s, args && (args->flags & 1u));
    });
}

int lj_render(lj_scene *scene, const LjRenderArgs *args, float *rgb_host) {
    return lj::guard([&]() {
        if (!scene || !rgb_host) throw LjError(LJ_ERR_INVALID_ARG, "lj_render: null argument");
        lj_context *ctx = scene->ctx;
        set_device(ctx);
        const size_t fb = (size_t)scene->flat.cam.width * scene->flat.cam.height * 3 * sizeof(float);
        if (ctx->frame.bytes < fb) ctx->frame.alloc(fb);
        RenderPlan plan = make_plan(scene, args);
        HIP_CHECK(hipMemsetAsync(ctx->frame.p, 0, fb, ctx->stream));
        run_render(scene, plan, (float *)ctx->frame.p, nullptr, ctx->stream, args && (args->flags & 1u));
        HIP_CHECK(hipMemcpyAsync(rgb_host, ctx->frame.p, fb, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int lj_render_samples(lj_scene *scene, const LjRenderArgs *args, float *radiance_host) {
    return lj::guard([&]() {
        if (!scene || !radiance_host || !args) throw LjError(LJ_ERR_INVALID_ARG, "lj_render_samples: null argument");
        if (!(args->crop_x1 > args->crop_x0 && args->crop_y1 > args->crop_y0)) throw LjError(LJ_ERR_INVALID_ARG, "lj_render_samples needs a crop window");
        if (args->world_size > 1) throw LjError(LJ_ERR_INVALID_ARG, "lj_render_samples is single-rank");
        set_device(scene->ctx);
        RenderPlan plan = make_plan(scene, args);
        run_render(scene, plan, nullptr, radiance_host, scene->ctx->stream, false);
    });
}

static int trace_batch(lj_scene *scene, int64_t n, const LjRay *rays_host, LjHit *hits_host, uint8_t *occ_host) {
    return lj::guard([&]() {
        if (!scene || !rays_host || n < 0) throw LjError(LJ_ERR_INVALID_ARG, "trace: bad argument");
        if (n == 0) return;
        lj_context *ctx = scene->ctx;
        set_device(ctx);
        DevBuf rays, out;
        rays.alloc((size_t)n * sizeof(LjRay));
        out.alloc((size_t)n * (hits_host ? sizeof(LjHit) : 1));
        HIP_CHECK(hipMemcpyAsync(rays.p, rays_host, (size_t)n * sizeof(LjRay), hipMemcpyHostToDevice, ctx->stream));
        int grid = (int)std::min<int64_t>((n + 255) / 256, ctx->n_cus * 4);
        ljd::launch_trace_rays(scene->dscene, rays.p, n, hits_host ? out.p : nullptr, hits_host ? nullptr : (unsigned char *)out.p, scene->small, grid, ctx->stream);
        HIP_CHECK(hipGetLastError());
        if (hits_host) HIP_CHECK(hipMemcpyAsync(hits_host, out.p, (size_t)n * sizeof(LjHit), hipMemcpyDeviceToHost, ctx->stream));
        else HIP_CHECK(hipMemcpyAsync(occ_host, out.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}
int lj_intersect(lj_scene *scene, int64_t n, const LjRay *rays_host, LjHit *hits_host) {
    if (!hits_host) { lj::set_last_error("lj_intersect: null output"); return LJ_ERR_INVALID_ARG; }
    return trace_batch(scene, n, rays_host, hits_host, nullptr);
}
int lj_occluded(lj_scene *scene, int64_t n, const LjRay *rays_host, uint8_t *occluded_host) {
    if (!occluded_host) { lj::set_last_error("lj_occluded: null output"); return LJ_ERR_INVALID_ARG; }
    return trace_batch(scene, n, rays_host, nullptr, occluded_host);
}

int lj_get_stats(const lj_scene *scene, LjStats *out) {
    if (!scene || !out) { lj::set_last_error("lj_get_stats: null argument"); return LJ_ERR_INVALID_ARG; }
    *out = scene->stats;
    return LJ_OK;
}

int lj_scene_info(const lj_scene *scene, LjSceneInfo *out) {
    if (!scene || !out) { lj::set_last_error("lj_scene_info: null argument"); return LJ_ERR_INVALID_ARG; }
    const lj::FlatScene &F = scene->flat;
    LjSceneInfo i{};
    i.width = F.cam.width; i.height = F.cam.height; i.spp = F.spp; i.max_depth = F.max_depth; i.rr_depth = F.rr_depth; i.integrator = F.integrator;
    i.n_triangles = F.n_triangles; i.n_spheres = F.n_spheres; i.n_bvh_nodes = (int64_t)F.nodes.size();
    i.bounds_radius = F.bounds_radius; for (int k = 0; k < 3; k++) i.bounds_center[k] = F.bounds_center[k];
    i.shadow_epsilon = F.shadow_epsilon;
    *out = i;
    return LJ_OK;
}

} // extern "C"
