// gfx950 volumetric path tracer (vol_path_tracing.h:503-869): persistent waves, one path per lane, regeneration at
// bounce granularity (dvol.h: vol_path_begin / vol_path_step); closest hits come from the BVH4 traversal steps of dtrav.h.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <algorithm>
#include <type_traits>
#include "dtrav.h"
#include "dvol.h"

namespace ljd {

// ---------------------------------------------------------------- the tracer's ray casts: lane by lane through the BVH4
// LJ_VOLPATH_STATS (a developer build, tools/dev/volpath_stats.sh): wave-level counts of how often each part of the volumetric tracer
// runs and how many lanes are active in it — slots: 0 traversal node iterations, 1 leaf steps, 2 closest() calls, 3 / 4 tracking iterations
// of the bounce loop / of shadow segments, 5 shadow segments, 6 vol_path_step calls.  64-bit words 2 + 2 s and 3 + 2 s of `counters`: wave-level events, lane-events.
#ifndef LJ_VOLPATH_STATS
#define LJ_VOLPATH_STATS 0
#endif
// SPHERES: 0 the scene holds no sphere; 1 spheres are tested where the traversal meets their leaves; 2 (a scene with a single sphere)
// the traversal passes over them — a sphere's leaf-ordered record has three zero vertices, which the triangle test rejects — and every
// ray tests every sphere afterwards.  Why: inlined into the lane-by-lane leaf step, the reference's double-precision sphere callback holds
// ~50 VGPRs at the tracer's register peak (65 - 112 spilled registers instead of 8 - 63); after the traversal its registers are free.
// The closest hit is the minimum of (t, primitive id) over everything tested, so where a test happens cannot change it.
template <int SPHERES>
struct DevTracer {
    const TreeView &tv;
    const DSphere *spheres; int n_spheres;
#if LJ_VOLPATH_STATS
    uint32_t ev[8], ln[8];   // ln: events this lane was active in; ev: events this lane was the first active lane of (their sum over a wave = the wave's events)
    __device__ __forceinline__ void tick(int s) {
        const unsigned long long b = __ballot(true);
        ln[s]++;
        if (__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u)) == 0u) ev[s]++;
    }
#else
    __device__ __forceinline__ void tick(int) {}
#endif
    __device__ __forceinline__ bool closest(f3 org, f3 dir, float tnear, float tfar, float &t, float &u, float &v, int &gprim) {
        LaneTrav L;
        L.ray.ox = org.x; L.ray.oy = org.y; L.ray.oz = org.z; L.ray.dx = dir.x; L.ray.dy = dir.y; L.ray.dz = dir.z;
        trav_begin(L, tnear, tfar);
        tick(2);
        while (L.cur != kDone) {
            while (L.cur >= 0 && L.cur != kDone) { tick(0); trav_node_step<false>(tv, L); }
            if (L.cur < 0) { tick(1); trav_leaf_step<false, SPHERES == 1>(tv, L, false); }
        }
        trav_finish(L);
        t = L.best.t; u = L.best.u; v = L.best.v; gprim = L.best.gprim;
        if (SPHERES == 2) {
            RayF ray; ray.ox = org.x; ray.oy = org.y; ray.oz = org.z; ray.dx = dir.x; ray.dy = dir.y; ray.dz = dir.z; ray.tnear = tnear; ray.tfar = tfar;
            for (int s = 0; s < n_spheres; s++) {
                double td;
                if (sphere_test(ray, spheres[s], td)) {
                    const float tf = (float)td; const int g = spheres[s].gprim;
                    if (tf < t || (tf == t && (gprim < 0 || g < gprim))) { t = tf; u = 0.0f; v = 0.0f; gprim = g; }   // the rule of trav_leaf_step on (t, gprim)
                }
            }
        }
        return gprim >= 0;
    }
};

// Persistent waves with path regeneration (the structure of k_mega): every lane carries one path, bounce by bounce (dvol.h
// vol_path_step); a lane whose path has ended takes the next camera sample of its wave's open range, and a wave whose range is used
// up takes the next `grab` samples off one grid-wide counter — so the lanes of a wave stay busy whatever the lengths of their paths
// (one whole path per lane left a wave waiting for its longest path).  A sample's value depends on its pcg32 stream only.
// counters[0..1]: bounce iterations (64 bit); counters[2]: the sample counter (zeroed before the launch).
template <class Ft, int SPHERES>
__device__ __forceinline__ void volpath_body(const DScene &sc, const DPass &pass, uint32_t n_samples, uint32_t grab, uint32_t *counters, int stack, int lds_nodes, int lds_prims, int *spill) {
    const TreeView tv = stage_tree(sc, stack, lds_nodes, lds_prims, spill, gridDim.x * kBlock, blockIdx.x * kBlock + threadIdx.x);
    DevTracer<SPHERES> tr{tv, sc.spheres, sc.n_spheres};
#if LJ_VOLPATH_STATS
    for (int k = 0; k < 8; k++) { tr.ev[k] = 0; tr.ln[k] = 0; }
#endif
    uint32_t *sample_counter = counters + 2;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t bounces = 0;
    bool live = false, exhausted = false;
    uint32_t w_next = 0, w_end = 0;   // the wave's open range of camera samples (wave-uniform)
    uint32_t sample = 0;
    VolPath P;
    auto finish = [&](uint32_t s, f3 rad, uint32_t nb) {
        if (!(isfinite(rad.x) && isfinite(rad.y) && isfinite(rad.z))) rad = mk3(0, 0, 0);   // render.cpp:138-141: a non-finite sample is left out
        float *o = pass.sample_rgb + 3ull * s;
        o[0] = rad.x; o[1] = rad.y; o[2] = rad.z;
        bounces += nb;
    };
    for (;;) {
        const unsigned long long dead = __ballot(!live);
        if (dead != 0ull && !exhausted) {
            if (w_next == w_end) {
                uint32_t b = 0;
                if (lane == 0u) b = atomicAdd(sample_counter, grab);
                b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                if (b >= n_samples) exhausted = true;
                else { w_next = b; w_end = (n_samples - b < grab) ? n_samples : b + grab; }
            }
            if (!exhausted) {
                const uint32_t left = w_end - w_next, n_dead = (uint32_t)__popcll(dead);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(dead >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dead, 0u));
                if (!live && rank < left) {
                    sample = w_next + rank;
                    const uint32_t p = fast_div(sample, pass.by_spp), k = sample - p * pass.spp;
                    const uint32_t pixel = pass.pixel_list[p];
                    f3 rad;
                    live = vol_path_begin<Ft>(sc, tr, (int)(pixel - fast_div(pixel, pass.by_width) * (uint32_t)sc.cam.width), (int)fast_div(pixel, pass.by_width), (uint64_t)pixel * pass.spp + k, pass.seed, P, rad);
                    if (!live) finish(sample, rad, 0u);   // (the single-shot estimators of version 1 and 2)
                }
                w_next += n_dead < left ? n_dead : left;
            }
        }
        if (__ballot(live) == 0ull) { if (exhausted) break; else continue; }
        if (live) {
            f3 rad;
            tr.tick(6);
            if (!vol_path_step<Ft>(sc, tr, P, rad)) { finish(sample, rad, P.bounce_iterations); live = false; }
        }
    }
    const uint32_t wb = wave_sum(bounces);
    if (lane == 0u && wb) atomicAdd((unsigned long long *)counters, (unsigned long long)wb);
#if LJ_VOLPATH_STATS
    for (int k = 0; k < 8; k++) {
        const uint32_t e = wave_sum(tr.ev[k]), l = wave_sum(tr.ln[k]);
        if (lane == 0u) { atomicAdd((unsigned long long *)counters + 2 + 2 * k, (unsigned long long)e); atomicAdd((unsigned long long *)counters + 3 + 2 * k, (unsigned long long)l); }
    }
#endif
}
// Instantiated per feature set of the scene (dshade.h: a scene of diffuse surfaces does not carry nine BSDFs), built for three waves per SIMD.
// (The tracker's own state, not the BSDFs, is what fills the registers: 211 VGPRs unconstrained for diffuse-only against 224 for everything.)
template <class Ft, int OCC, int SPHERES>
__global__ void __launch_bounds__(kBlock, OCC) k_volpath(DScene sc, DPass pass, uint32_t n_samples, uint32_t grab, uint32_t *counters, int stack, int lds_nodes, int lds_prims, int *spill) {
    volpath_body<Ft, SPHERES>(sc, pass, n_samples, grab, counters, stack, lds_nodes, lds_prims, spill);
}

// ---------------------------------------------------------------- launcher
void launch_volpath(const DScene &sc, const DPass &pass, uint32_t n_samples, uint32_t *counters, const ExtendConfig &cfg, int shade_variant, bool plain, int *spill, int grid, hipStream_t s) {
    if (!n_samples) return;
    // three waves per SIMD (168 VGPRs, 65 - 115 of them spilled) for every scene: with regeneration it beats the unconstrained two-wave build
    // on homogeneous and heterogeneous media alike (hetvol 155 vs 187 ms, vol_cbox_teapot 180 vs 235, profiles/r03_sweeps.txt); the two-wave
    // build of the all-features set stays selectable (LJ_TUNE_VOLPATH_OCC=2)
    int occ = 3;
    if (const char *e = getenv("LJ_TUNE_VOLPATH_OCC")) occ = atoi(e);
    // camera samples a wave takes off the counter at a time: every wave draws ~16 times, at least four samples per lane and at most 128 —
    // a single address takes ~90 atomics per microsecond, which short paths (an absorbing medium: one step per sample) would otherwise feel
    const uint64_t waves = (uint64_t)grid * (kBlock / 64);
    uint32_t grab = (uint32_t)std::min<uint64_t>(8192, std::max<uint64_t>(256, (n_samples / (waves * 16)) & ~(uint64_t)63));
    if (const char *e = getenv("LJ_TUNE_VOLPATH_GRAB")) grab = (uint32_t)(atoi(e) < 64 ? 64 : atoi(e)) & ~63u;
    auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), cfg.smem, s, sc, pass, n_samples, grab, counters, cfg.stack, cfg.lds_nodes, cfg.lds_prims, spill); };
    // feature sets compiled for this kernel: diffuse only / the three classic materials / everything (the smallest that covers the scene's)
    const int v = shade_variant <= 0 ? 0 : (shade_variant <= 3 ? 3 : kShadeVariantAll);
    // 0: no spheres; 2: a handful, tested after the traversal; 1: tested inside it (see DevTracer)
    int sph = cfg.spheres == 0 ? 0 : (sc.n_spheres == 1 ? 2 : 1);   // (measured: one sphere — hetvol 184 -> 192 Msamples/s; two or three — volpath_test4 / 5 / 6 7 - 9 % slower than inside the traversal)
    if (const char *e = getenv("LJ_TUNE_VOLPATH_SPHERES")) sph = cfg.spheres == 0 ? 0 : (atoi(e) == 1 ? 1 : sph);
    auto pick = [&](auto ft, auto occ_c) {
        using Ft = decltype(ft); constexpr int O = decltype(occ_c)::value;
        if (sph == 0) launch(k_volpath<Ft, O, 0>); else if (sph == 2) launch(k_volpath<Ft, O, 2>); else launch(k_volpath<Ft, O, 1>);
    };
    // `plain`: the classic materials in constant colours under mesh lights (vol_cbox_teapot: diffuse walls, a rough dielectric teapot) — without the
    // texture, environment-map and sphere-light code the tracer spills fewer registers at three waves per SIMD
    using FeatClassicPlain = ShadeFeat<0x007u, false, false, false>;
    if (v == 0) pick(FeatLambert{}, std::integral_constant<int, 3>{});
    else if (v == 3 && plain) pick(FeatClassicPlain{}, std::integral_constant<int, 3>{});
    else if (v == 3) pick(FeatClassic{}, std::integral_constant<int, 3>{});
    else if (occ <= 2) launch(k_volpath<FeatAll, 2, 1>);
    else pick(FeatAll{}, std::integral_constant<int, 3>{});
}
int volpath_blocks_per_cu(const DScene &sc) {   // workgroups that stay resident per CU: the persistent grid is n_cus x this
    if (const char *e = getenv("LJ_TUNE_VOLPATH_BLOCKS_PER_CU")) return atoi(e) > 0 ? atoi(e) : 1;
    int occ = 3;
    if (const char *e = getenv("LJ_TUNE_VOLPATH_OCC")) occ = atoi(e);
    return occ < 2 ? 2 : (occ > 3 ? 3 : occ);
}

} // namespace ljd
