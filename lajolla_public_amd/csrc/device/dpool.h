// Pieces shared by the extend kernels of kernels.hip (BVH4, trees inside the LDS image) and extend8.hip (BVH8, trees beyond it): queue
// record helper, the batched-query records, and the pooled leaf phase — a wave lists its (ray, primitive) pairs in LDS and tests them 64 at
// a time on whichever lane is free.
#pragma once
#include <hip/hip_runtime.h>
#include "dstage.h"
#include "dtrace.h"
#include "dconfig.h"

namespace ljd {

__device__ __forceinline__ Rec4 mk4(float x, float y, float z, float w) { Rec4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

constexpr uint32_t kChunk = 256;   // queue slots a wave draws at a time (segments are multiples of it)

// batched ray queries (lj_intersect / lj_occluded)
struct RayIO { float org[3]; float tnear; float dir[3]; float tfar; };
struct HitIO { float t, u, v; int32_t shape_id, prim_id; };

// ---- the leaf phase of k_extend, pooled.  In the while-while loop the lanes that sit on a leaf hold 1 ... 8 primitives each and the
// others none: tested lane by lane, a primitive round runs at ~30 % lane occupancy (sponza 27 %, disney_bsdf 35 %).  Here the wave lists
// its (ray, primitive) pairs in LDS and tests them 64 at a time on whichever lane is free; the ray of a pair comes from its owner's
// registers (ds_bpermute), results are merged per owner by one 64-bit LDS minimum on (t, global primitive id) — the same order the
// lane-by-lane test applies, so the hit record is the same, bit for bit — and the winner leaves its unnormalised barycentrics beside it.
#ifndef LJ_EXT_POOL
#define LJ_EXT_POOL 1
#endif
constexpr uint32_t kPoolCap = 256;                                      // pairs listed at a time (a wave holds at most 64 x 16: a held leaf and the one a lane sits on)
constexpr uint32_t kWavePoolBytes = 64 * 8 + 64 * 16 + kPoolCap * 4;    // keys | winners (U, V, S, t) | items (owner lane | leaf-order primitive index << 6)
struct LeafPool { LJ_LDS unsigned long long *keys; LJ_LDS v4f *win; LJ_LDS uint32_t *items; };

__device__ __forceinline__ LeafPool leaf_pool_at(uint32_t at) {
    LJ_LDS char *w = (LJ_LDS char *)lj_smem + at + (threadIdx.x >> 6) * kWavePoolBytes;
    LeafPool lp;
    lp.keys = (LJ_LDS unsigned long long *)w; lp.win = (LJ_LDS v4f *)(w + 64 * 8); lp.items = (LJ_LDS uint32_t *)(w + 64 * 8 + 64 * 16);
    lp.keys[threadIdx.x & 63u] = ~0ull;
    return lp;
}
__device__ __forceinline__ float lane_read(int src4, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(v))); }

// Tests the `n_items` listed (owner lane, primitive) pairs of the wave, 64 at a time; called by the whole wave (the bpermutes need every owner
// lane active).  The view `tv` says where primitives live (gprims, and lprims for the first n_lprims when the scene is staged in LDS).
template <bool RESIDENT, bool SPHERES, class View, class Lane>
__device__ __forceinline__ void pool_test_items(const View &tv, const LeafPool &lp, const Lane &L, const uint32_t n_items, uint32_t &rounds) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t r = 0; r < n_items; r += 64u, rounds++) {
        const bool act = r + lane < n_items;
        const uint32_t it = act ? lp.items[r + lane] : lane;
        const uint32_t src = it & 63u;
        const int src4 = (int)(src << 2);
        RayF ray;
        ray.ox = lane_read(src4, L.ray.ox); ray.oy = lane_read(src4, L.ray.oy); ray.oz = lane_read(src4, L.ray.oz);
        ray.dx = lane_read(src4, L.ray.dx); ray.dy = lane_read(src4, L.ray.dy); ray.dz = lane_read(src4, L.ray.dz);
        ray.tnear = lane_read(src4, L.ray.tnear);
        ray.tfar = SPHERES ? lane_read(src4, L.ray.tfar) : 0.0f;
        const float tbest = lane_read(src4, L.best.t);
        const int pi = (int)(it >> 6);
        bool hit = false;
        unsigned long long key = 0ull;
        v4f w; w.x = 0.0f; w.y = 0.0f; w.z = 1.0f; w.w = 0.0f;
        if (act) {
            v4f p0, p1, p2;
            if (RESIDENT || pi < tv.n_lprims) { const int S = tv.prim_stride; p0 = tv.lprims[pi]; p1 = tv.lprims[S + pi]; p2 = tv.lprims[2 * S + pi]; }
            else { p0 = tv.gprims[3 * pi]; p1 = tv.gprims[3 * pi + 1]; p2 = tv.gprims[3 * pi + 2]; }
            const int gprim = __float_as_int(p0.w);
            if (!SPHERES || __float_as_int(p1.w) == 0) {
                const float v0[3] = {p0.x, p0.y, p0.z}, v1[3] = {p1.x, p1.y, p1.z}, v2[3] = {p2.x, p2.y, p2.z};
                float t = 0.0f, U = 0.0f, V = 0.0f, S = 1.0f;
                hit = tri_test_raw(ray, tbest, v0, v1, v2, t, U, V, S);   // t > tnear >= 0: its bits order like the value
                w.x = U; w.y = V; w.z = S; w.w = t;
                key = ((unsigned long long)f2u(t) << 32) | (unsigned long long)(uint32_t)gprim;
            } else {
                double td = 0.0;
                hit = sphere_test(ray, tv.spheres[__float_as_int(p2.w)], td);
                const float t = (float)td;                                // t >= tnear >= 0; a zero of either sign orders as +0
                w.w = t;
                key = ((unsigned long long)(t == 0.0f ? 0u : f2u(t)) << 32) | (unsigned long long)(uint32_t)gprim;
            }
            if (hit) (void)__hip_atomic_fetch_min(&lp.keys[src], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        // (a wave's LDS operations complete in order: every pair's minimum has landed when the read below is issued)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (hit && __hip_atomic_load(&lp.keys[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == key) lp.win[src] = w;   // one winner per ray: a leaf holds a primitive once
    }
}
// A ray's pooled result, if any: the rule of trav_leaf_step on (t, gprim).  Returns true when an any-hit ray is finished.
template <class Lane>
__device__ __forceinline__ bool pool_collect(const LeafPool &lp, Lane &L, const bool any_hit) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long key = lp.keys[lane];
    if (key == ~0ull) return false;
    lp.keys[lane] = ~0ull;
    const int gprim = (int)(uint32_t)key;
    const v4f w = lp.win[lane];
    // (selects, as in trav_leaf_step: an any-hit ray only needs `gprim`)
    const bool take = any_hit | (w.w < L.best.t) | ((w.w == L.best.t) & ((L.best.gprim < 0) | (gprim < L.best.gprim)));
    L.best.t = take ? w.w : L.best.t; L.best.u = take ? w.x : L.best.u; L.best.v = take ? w.y : L.best.v;
    L.best_S = take ? w.z : L.best_S; L.best.gprim = take ? gprim : L.best.gprim;
    return any_hit;
}


// extend8.hip
size_t extend8_smem(const ExtendConfig &cfg);
void launch_extend8(const DScene &sc, const DQueue &q, const DBlockState *blocks, uint32_t grid, uint32_t seg, uint32_t *work, const uint32_t *chunk_list, uint32_t parity, const ExtendConfig &cfg, int *spill, unsigned long long *stats, hipStream_t s);
void launch_trace_rays8(const DScene &sc, const void *rays, long long n, void *hits, unsigned char *occ, const ExtendConfig &cfg, int *spill, int grid, hipStream_t s);

} // namespace ljd
