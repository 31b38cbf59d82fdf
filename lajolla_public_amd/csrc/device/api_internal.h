// Internals shared by the C-ABI translation units that own the GPU (api_device.hip: context, upload, render; queries.hip:
// the batched per-object queries): HIP error boundary, device buffers, the context and scene objects behind the opaque
// handles of include/lajolla_hip.h, and the kernel launchers of kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "../host/api_common.h"
#include "../host/flatten.h"
#include "dtypes.h"
#include "dconfig.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace ljd {
ShadeConfig shade_config(size_t n_prims, size_t n_materials, size_t n_lights, size_t n_light_tris, size_t n_light_tri_cdf, size_t n_images3, size_t n_images1, size_t n_env_marg);
int shade_variant(uint32_t kinds, bool textured, bool envmap, bool sphere_lights);
ExtendConfig extend_config(int n_nodes, int n_prims, int bvh_depth, int n_spheres, int n_nodes8, int bvh8_depth, bool open_scene);
int max_stack_depth();
void launch_extend(const DScene &sc, const DQueue &q, const DBlockState *blocks, uint32_t grid, uint32_t seg, uint32_t *work, const uint32_t *chunk_list, uint32_t parity, const ExtendConfig &cfg, int *spill, unsigned long long *stats, hipStream_t s);
bool shade_sorts_segments(const ShadeConfig &cfg);
void launch_shade(const DScene &sc, const DPass &pass, const DQueue &q, const DQueue &qo, uint32_t *sort_perm, uint8_t *sort_keys, DBlockState *blocks, uint32_t n_blocks, uint32_t seg, const ShadeConfig &cfg, uint32_t *work, uint32_t *chunk_list, uint32_t parity, uint32_t extend_waves, hipStream_t s);
void launch_resolve(const DPass &pass, uint32_t n_pixels, float *rgb, hipStream_t s);
size_t tail_smem(const ExtendConfig &ecfg, const ShadeConfig &scfg);
void launch_tail(const DScene &sc, const DPass &pass, const DQueue &q, DBlockState *blocks, uint32_t n_blocks, uint32_t seg, const ExtendConfig &ecfg, const ShadeConfig &scfg, int *spill, hipStream_t s);
void launch_aux(const DScene &sc, const uint32_t *pixel_list, uint32_t n_pixels, int integrator, float *rgb, const ExtendConfig &cfg, int *spill, int grid, hipStream_t s);
void launch_volpath(const DScene &sc, const DPass &pass, uint32_t n_samples, uint32_t *counters, const ExtendConfig &cfg, int shade_variant, bool plain, int *spill, int grid, hipStream_t s);
int volpath_blocks_per_cu(const DScene &sc);
void launch_trace_rays(const DScene &sc, const void *rays, long long n, void *hits, unsigned char *occ, const ExtendConfig &cfg, int *spill, int grid, hipStream_t s);
// mega.hip
size_t mega_smem(const DScene &sc, const ShadeConfig &scfg);
int mega_blocks_per_cu(const ShadeConfig &scfg);
void launch_mega(const DScene &sc, const DPass &pass, const ShadeConfig &scfg, bool spheres, uint32_t n_samples, uint32_t grab, uint32_t *sample_counter, unsigned long long *stats, int grid, hipStream_t s);
void launch_trace_rays_scan(const DScene &sc, const void *rays, long long n, void *hits, unsigned char *occ, int grid, hipStream_t s);
}

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

template <typename T> inline void upload(DevBuf &b, const std::vector<T> &v, hipStream_t s) {
    b.alloc(((std::max<size_t>(v.size(), 1) * sizeof(T)) + 31) & ~(size_t)15);
    if (!v.empty()) HIP_CHECK(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

static constexpr size_t kQueueSlotBytes = 128;  // eight 16-byte records per path
static constexpr uint32_t kMaxBlocks = 8192;

static constexpr uint32_t kMaxLanes = 8;   // render lanes (streams) a context can run side by side
struct lj_context {
    int device = 0;
    // lifetime: a context outlives its scenes whatever order the caller destroys them in — lj_context_destroy on a context that still
    // has scenes only marks it, and the last lj_scene_destroy releases it
    int live_scenes = 0; bool doomed = false;
    hipStream_t stream = nullptr;
    hipStream_t lane_streams[kMaxLanes - 1] = {};   // further lanes of a render (run_render)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_caller = nullptr;
    int n_cus = 256;
    // workspace, grown on demand and reused across renders
    DevBuf queue_mem; uint32_t queue_capacity = 0;
    DevBuf queue_mem2, sort_perm, sort_keys; uint32_t queue2_capacity = 0;   // segment-sorted shading: the second record set, slot permutation, keys
    DevBuf chunk_counter, chunk_list;  // the extend kernel's work counters and the two lists of live queue chunks
    DevBuf spill;  // overflow levels of the traversal stacks: spill_levels x (grid * 256) ints
    DevBuf blocks, sample_rgb, pixel_list, frame;
    uint64_t pixel_list_key = 0;   // which pixel list `pixel_list` holds (RenderPlan::pixels_key), 0: none
    DevBuf mega_state;   // k_mega: [0] the grid-wide camera-sample counter (uint32), [8..] five 64-bit statistics
    ljd::DBlockState *blocks_host = nullptr;  // pinned, kMaxBlocks entries
    unsigned long long *stats_host = nullptr; // pinned, 8 entries: where a launch's statistics are read back (one wait per pass, no staged copy)
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
};

struct lj_scene {
    lj_context *ctx = nullptr;
    lj::FlatScene flat;  // host copy (tables for lj_scene_info; arrays already uploaded)
    DevBuf nodes, nodes8, leaf_prims, prims, spheres, materials, lights, light_cdf, light_tris, light_tri_cdf, images3, images1, texels, env_tables;
    DevBuf media, volume_data, shape_media, scan_leaves;
    ljd::DScene dscene{};
    ljd::ExtendConfig ecfg{};
    ljd::ShadeConfig scfg{};
    uint32_t feat_kinds = 0; bool feat_textured = false, feat_envmap = false, feat_sphere_lights = false;   // what the scene holds (picks scfg.variant)
    std::shared_ptr<const std::vector<uint32_t>> plan_pixels; uint64_t plan_pixels_key = 0;   // the last render plan's pixel list (make_plan)
    std::vector<int64_t> shape_first_gprim;   // global primitive id of each shape's first primitive (shape order, then triangle order)
    LjStats stats{};
};

