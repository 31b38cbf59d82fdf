// Multi-device rendering from ONE process (include/lajolla_hip.h, "device groups"): the native replacement of the reference's tile
// loop across GPUs (render.cpp:71-101 over parallel.cpp:183-237).  A group owns one context per device and — for two or more
// distinct devices — an RCCL communicator (ncclCommInitAll); a group scene is the scene uploaded to every device.  lj_group_render
// deals the image's 16x16 tiles round-robin to the devices (tile t goes to device t mod N), renders every share concurrently (one
// host thread per device drives its wavefront loop), sum-reduces the N float frames onto device 0 with one ncclReduce on the render
// streams, and copies the frame to the host.  Every pixel has exactly one non-zero contributor, so the sum is exact and the image is
// bit-identical to a one-device render.
//
// RCCL is bound at run time (dlopen of librccl.so on first use by a group of >= 2 distinct devices), so the library keeps loading on
// a machine without it and a single-device user never touches it.  A group whose ids repeat a device ("logical ranks": how the N-rank
// path is exercised on a one-GPU box) cannot form an RCCL communicator — RCCL refuses duplicate devices — and sums its frames with
// a device kernel instead; LJ_GROUP_NO_RCCL=1 forces that path for distinct devices too (peer copies).
#include "api_internal.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <set>
#include <thread>

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl &rccl() {
    static Rccl r;
    if (r.lib) return r;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
    if (!r.lib) throw LjError(LJ_ERR_DEVICE, std::string("a group of several devices needs RCCL and librccl.so could not be loaded: ") + dlerror());
    auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p) throw LjError(LJ_ERR_DEVICE, std::string("librccl.so lacks ") + n); return p; };
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll"); r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.Reduce = (decltype(r.Reduce))sym("ncclReduce"); r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd"); r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    return r;
}

#define RCCL_CHECK(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) \
    throw LjError(LJ_ERR_DEVICE, std::string(#expr) + " failed: " + rccl().GetErrorString(_r)); } while (0)

__global__ void k_add_frame(float *dst, const float *src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

} // namespace

struct lj_device_group {
    std::vector<lj_context *> ctx;
    std::vector<ncclComm_t> comm;   // one per device when RCCL is in use, else empty
    bool use_rccl = false;
    int live_scenes = 0; bool doomed = false;   // (lifetime: as lj_context — the group is released by its last scene if it was destroyed first)
    std::vector<DevBuf *> frame;    // one full float frame per device, grown on demand
    DevBuf staging;                 // device 0: a peer's frame, for the RCCL-free sum
    ~lj_device_group() {
        for (size_t i = 0; i < comm.size(); i++) if (comm[i]) (void)rccl().CommDestroy(comm[i]);
        for (size_t i = 0; i < ctx.size(); i++) { if (ctx[i]) (void)hipSetDevice(ctx[i]->device); delete frame[i]; }
        if (!ctx.empty() && ctx[0]) { (void)hipSetDevice(ctx[0]->device); staging.release(); }
        for (lj_context *c : ctx) lj_context_destroy(c);
    }
};

struct lj_group_scene {
    lj_device_group *group = nullptr;
    std::vector<lj_scene *> scene;
    LjStats stats{};
    ~lj_group_scene() { for (lj_scene *s : scene) lj_scene_destroy(s); }   // (lj_group_scene_destroy does the group's bookkeeping)
};

extern "C" {

int lj_group_create(int n_devices, const int *device_ids, lj_device_group **out) {
    return lj::guard([&]() {
        if (!out || n_devices <= 0 || n_devices > 64) throw LjError(LJ_ERR_INVALID_ARG, "lj_group_create: need 1..64 devices and an output pointer");
        *out = nullptr;
        auto g = std::make_unique<lj_device_group>();
        std::vector<int> ids(n_devices);
        for (int i = 0; i < n_devices; i++) ids[i] = device_ids ? device_ids[i] : i;
        for (int i = 0; i < n_devices; i++) {
            lj_context *c = nullptr;
            g->ctx.push_back(nullptr); g->frame.push_back(new DevBuf());
            const int rc = lj_context_create(ids[i], &c);
            if (rc != LJ_OK) throw LjError(rc, lj_last_error());
            g->ctx[i] = c;
        }
        const bool distinct = std::set<int>(ids.begin(), ids.end()).size() == (size_t)n_devices;
        const bool off = getenv("LJ_GROUP_NO_RCCL") && atoi(getenv("LJ_GROUP_NO_RCCL")) != 0;
        // (LJ_GROUP_FORCE_RCCL=1: a communicator even for a single device — the one-rank self-test of the RCCL binding on a one-GPU box)
        const bool force = getenv("LJ_GROUP_FORCE_RCCL") && atoi(getenv("LJ_GROUP_FORCE_RCCL")) != 0;
        if ((n_devices > 1 || force) && distinct && !off) {
            g->comm.assign(n_devices, nullptr);
            RCCL_CHECK(rccl().CommInitAll(g->comm.data(), n_devices, ids.data()));
            g->use_rccl = true;
        }
        *out = g.release();
    });
}

void lj_group_destroy(lj_device_group *group) {
    if (!group) return;
    if (group->live_scenes > 0) { group->doomed = true; return; }   // released by the last lj_group_scene_destroy
    delete group;
}
int lj_group_size(const lj_device_group *group) { return group ? (int)group->ctx.size() : 0; }
lj_context *lj_group_context(lj_device_group *group, int i) { return (group && i >= 0 && i < (int)group->ctx.size()) ? group->ctx[i] : nullptr; }
int lj_group_uses_rccl(const lj_device_group *group) { return (group && group->use_rccl) ? 1 : 0; }

int lj_group_scene_upload(lj_device_group *group, const LjSceneDesc *desc, lj_group_scene **out) {
    return lj::guard([&]() {
        if (!group || !desc || !out) throw LjError(LJ_ERR_INVALID_ARG, "lj_group_scene_upload: null argument");
        *out = nullptr;
        auto gs = std::make_unique<lj_group_scene>();
        gs->group = group;
        for (lj_context *c : group->ctx) {   // the scene is replicated: < 20 MB for the largest shipped one
            lj_scene *s = nullptr;
            const int rc = lj_scene_upload(c, desc, &s);
            if (rc != LJ_OK) throw LjError(rc, lj_last_error());
            gs->scene.push_back(s);
        }
        group->live_scenes++;
        *out = gs.release();
    });
}

void lj_group_scene_destroy(lj_group_scene *scene) {
    if (!scene) return;
    lj_device_group *g = scene->group;
    delete scene;
    if (--g->live_scenes == 0 && g->doomed) lj_group_destroy(g);
}
lj_scene *lj_group_scene_member(lj_group_scene *scene, int i) { return (scene && i >= 0 && i < (int)scene->scene.size()) ? scene->scene[i] : nullptr; }

int lj_group_render(lj_group_scene *gs, const LjRenderArgs *args, float *rgb_host) {
    return lj::guard([&]() {
        if (!gs || !rgb_host) throw LjError(LJ_ERR_INVALID_ARG, "lj_group_render: null argument");
        if (args && (args->world_size > 1 || args->rank != 0)) throw LjError(LJ_ERR_INVALID_ARG, "lj_group_render shards by itself: leave rank / world_size at 0");
        lj_device_group *g = gs->group;
        const int n = (int)g->ctx.size();
        const lj::FlatScene &F = gs->scene[0]->flat;
        const size_t count = (size_t)F.cam.width * F.cam.height * 3, fb = count * sizeof(float);
        for (int i = 0; i < n; i++) { HIP_CHECK(hipSetDevice(g->ctx[i]->device)); if (g->frame[i]->bytes < fb) g->frame[i]->alloc(fb); }
        // ---- every device renders its share; one host thread each (the wavefront loop of a context blocks its caller)
        std::vector<int> rc(n, LJ_OK); std::vector<std::string> msg(n);
        auto work = [&](int i) {
            LjRenderArgs a{};
            if (args) a = *args; else a.max_depth = INT32_MIN;
            a.rank = i; a.world_size = n;
            rc[i] = lj_render_device(gs->scene[i], &a, (float *)g->frame[i]->p, nullptr);
            if (rc[i] != LJ_OK) msg[i] = lj_last_error();
        };
        std::vector<std::thread> th;
        for (int i = 1; i < n; i++) th.emplace_back(work, i);
        work(0);
        for (auto &t : th) t.join();
        for (int i = 0; i < n; i++) if (rc[i] != LJ_OK) throw LjError(rc[i], "device " + std::to_string(g->ctx[i]->device) + " (share " + std::to_string(i) + "): " + msg[i]);
        // ---- one sum-reduce of the float frames onto device 0
        if (g->use_rccl) {
            RCCL_CHECK(rccl().GroupStart());
            try {
                for (int i = 0; i < n; i++) RCCL_CHECK(rccl().Reduce(g->frame[i]->p, g->frame[0]->p, count, ncclFloat, ncclSum, 0, g->comm[i], g->ctx[i]->stream));
            } catch (...) { (void)rccl().GroupEnd(); throw; }   // never leave the group call open
            RCCL_CHECK(rccl().GroupEnd());
        } else if (n > 1) {
            lj_context *c0 = g->ctx[0];
            HIP_CHECK(hipSetDevice(c0->device));
            for (int i = 1; i < n; i++) {
                const float *src = (const float *)g->frame[i]->p;
                if (g->ctx[i]->device != c0->device) {   // distinct devices without RCCL: stage the peer's frame on device 0
                    if (g->staging.bytes < fb) g->staging.alloc(fb);
                    HIP_CHECK(hipMemcpyPeerAsync(g->staging.p, c0->device, g->frame[i]->p, g->ctx[i]->device, fb, c0->stream));
                    src = (const float *)g->staging.p;
                }
                hipLaunchKernelGGL(k_add_frame, dim3(1024), dim3(256), 0, c0->stream, (float *)g->frame[0]->p, src, count);
                HIP_CHECK(hipGetLastError());
            }
        }
        HIP_CHECK(hipSetDevice(g->ctx[0]->device));
        HIP_CHECK(hipMemcpyAsync(rgb_host, g->frame[0]->p, fb, hipMemcpyDeviceToHost, g->ctx[0]->stream));
        HIP_CHECK(hipStreamSynchronize(g->ctx[0]->stream));
        // ---- statistics: sums over the shares; times are the slowest share's
        LjStats tot{};
        for (int i = 0; i < n; i++) {
            LjStats s{}; (void)lj_get_stats(gs->scene[i], &s);
            tot.samples += s.samples; tot.bounce_iterations += s.bounce_iterations; tot.rays_closest += s.rays_closest; tot.rays_shadow += s.rays_shadow;
            tot.queue_bytes += s.queue_bytes; tot.extend_launches += s.extend_launches; tot.shade_launches += s.shade_launches; tot.mega_launches += s.mega_launches;
            tot.extend_bytes += s.extend_bytes; tot.shade_bytes += s.shade_bytes; tot.mega_bytes += s.mega_bytes; tot.path_steps += s.path_steps;
            tot.wavefront_steps = std::max(tot.wavefront_steps, s.wavefront_steps); tot.render_ms = std::max(tot.render_ms, s.render_ms);
        }
        gs->stats = tot;
    });
}

int lj_group_get_stats(const lj_group_scene *scene, LjStats *out) {
    if (!scene || !out) { lj::set_last_error("lj_group_get_stats: null argument"); return LJ_ERR_INVALID_ARG; }
    *out = scene->stats;
    return LJ_OK;
}

} // extern "C"
