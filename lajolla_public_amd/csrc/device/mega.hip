// k_mega — the path tracer of a TINY scene (at most 32 BVH leaves / 256 primitives: cbox, veach_mi, the single-object test
// scenes) as ONE persistent launch: every lane carries one path in registers from its camera ray to its end and then takes
// the next camera sample off a grid-wide counter (path regeneration), so a wave stays full until the frame runs out of
// samples.  No path queue exists for such a scene: nothing but the finished per-sample radiance ever goes to HBM.
//
// Why a different plan for tiny scenes.  The wavefront kernels (kernels.hip) cut the bounce loop at its ray casts because a
// BVH traversal of unknown length and a shading step of unknown kind do not share a wave well.  A scene whose whole tree is
// a handful of leaves has neither problem: the closest hit is found by testing EVERY leaf box against the ray — a
// straight-line scan with the boxes in scalar registers, no stack, no node fetch, no lane idle (dscan below) — and the few
// (ray, leaf) candidates that survive are pooled per wave in LDS and tested by whichever lane is free, 64 at a time, with
// the closest hit merged by an LDS atomic min on (t, primitive id): the same total order the BVH traversal minimises, so
// hit records are bit-identical to k_extend's.  With the trace that regular, fusing it with shade_path (dshade.h — the very
// function k_shade runs) costs no occupancy and removes the queue round trip (290 B per path-step) altogether.
//
// The per-sample values are those of the wavefront kernels bit for bit: same shade_path, same primitive tests, same order
// of the radiance additions (path_tracing.h:207 before the next vertex's emission); a sample's value depends on its pcg32
// stream only, never on the lane, wave or launch that computed it.
#include "dstage.h"
#include "dtrace.h"
#include "dconfig.h"

namespace ljd {

#define LJ_CONST __attribute__((address_space(4)))   // constant address space: uniform loads become s_load (scalar cache)

constexpr uint32_t kItemCap = 512;                    // (ray, leaf) candidates pooled per wave and pass
constexpr uint32_t kWaveScanBytes = 64 * 48 + 64 * 8 + 64 * 4 + kItemCap * 2;   // rays | keys | occluded flags | items

struct ScanCtx {
    const LJ_CONST float *boxes;        // DScanLeaf records (8 dwords each), wave-uniform reads
    int n_used;                         // leaves in the table
    const LJ_LDS int *leaf_tab;         // (first, count) per leaf
    const LJ_LDS v4f *lprims; int prim_stride;   // leaf-ordered primitives, transposed: v4f k of primitive i at [k * stride + i]
    const DSphere *spheres;
    // this wave's scratch
    LJ_LDS v4f *rays;                   // [lane * 3 + {0: org | tfar_shadow, 1: dir_ext | tnear_ext, 2: dir_shadow | tnear_shadow}]; tfar_ext = inf
    LJ_LDS unsigned long long *keys;    // closest hit of lane's extension ray: float bits of t << 32 | gprim << 16 | leaf-order index
    LJ_LDS uint32_t *occl;              // != 0: lane's shadow ray is blocked
    LJ_LDS uint16_t *items;
};

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Leaf-box scan: one bit of a ray's mask per leaf box its segment [tnear, tfar] overlaps.  Same slab arithmetic as the BVH node step
// (t = plane * (1/d) - o * (1/d), v_rcp reciprocals, exit widened by 4 ulp; the boxes carry the builder's 1e-5 padding), with
// min / max instead of sign-selected planes because the planes are scalars here.  A direction component closer to zero than
// 1e-18 is moved there: the slab then spans |t| < 1e18 * (distance to the plane) instead of producing inf - inf.  It can only
// mis-decide a slab whose plane lies within float rounding of the ray's origin, and the padding keeps every primitive of the box
// 100 times further inside than that.
struct ScanRay { float ix, iy, iz, ox, oy, oz; };
__device__ __forceinline__ ScanRay scan_ray(f3 org, f3 dir) {
    const float tiny = 1e-18f;
    const float dx = fabsf(dir.x) < tiny ? copysignf(tiny, dir.x) : dir.x, dy = fabsf(dir.y) < tiny ? copysignf(tiny, dir.y) : dir.y,
                dz = fabsf(dir.z) < tiny ? copysignf(tiny, dir.z) : dir.z;
    ScanRay r;
    r.ix = __builtin_amdgcn_rcpf(dx); r.iy = __builtin_amdgcn_rcpf(dy); r.iz = __builtin_amdgcn_rcpf(dz);
    r.ox = org.x * r.ix; r.oy = org.y * r.iy; r.oz = org.z * r.iz;
    return r;
}
// One box against one ray; FAR: the segment has a far end (shadow rays; extension rays run to infinity).  Returns te - 1.0000005 tx in one
// rounding: NEGATIVE (sign bit set) when the segment overlaps the box.  The scan shifts that sign bit into the ray's candidate mask with
// one v_alignbit — no compare, no select.  (Against `te <= round(tx * 1.0000005)` the decision can differ only for |te - tx c| below one
// rounding, i.e. for boxes the exact ray touches in a single point behind its own 4-ulp allowance; which boxes are entered never changes
// a hit — the closest hit is the (t, primitive) minimum over every box that holds it — it only has to stay conservative.)
// `tnear` must be a canonical number (the callers pass max(tnear, 0)), so that the maximum below compiles without a quieting copy.
template <bool FAR>
__device__ __forceinline__ float scan_box(const float (&b)[6], const ScanRay &r, float tnear, float tfar) {
    const float ax = __builtin_fmaf(b[0], r.ix, -r.ox), bx = __builtin_fmaf(b[3], r.ix, -r.ox);
    const float ay = __builtin_fmaf(b[1], r.iy, -r.oy), by = __builtin_fmaf(b[4], r.iy, -r.oy);
    const float az = __builtin_fmaf(b[2], r.iz, -r.oz), bz = __builtin_fmaf(b[5], r.iz, -r.oz);
    const float te = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tnear));
    float tx = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    if (FAR) tx = fminf(tx, tfar);
    return __builtin_fmaf(tx, -1.0000005f, te);
}
__device__ __forceinline__ uint32_t shift_in_sign(uint32_t mask, float d) { return __builtin_amdgcn_alignbit(mask, f2u(d), 31u); }   // (mask << 1) | sign(d)

// Both rays of a path (E: some lane has an extension ray, S: some lane has a shadow ray — wave-uniform, decided outside the loop) against
// every leaf box in ONE pass over the table: the boxes are fetched (scalar loads, four boxes ahead) once, and the two independent slab
// chains interleave.  Bit (n_used - 1 - k) of a ray's mask = its segment overlaps leaf box k (the bits are shifted in from below).
template <bool E, bool S>
__device__ __forceinline__ void scan_leaf_boxes(const ScanCtx &sx, const ScanRay &re, float tnear_e, const ScanRay &rs, float tnear_s, float tfar_s, uint32_t &me, uint32_t &ms) {
    const int n = sx.n_used;
    int k = 0;
    for (; k + 4 <= n; k += 4) {
        float b[4][6];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int j = 0; j < 6; j++) b[c][j] = sx.boxes[(k + c) * 8 + j];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (E) me = shift_in_sign(me, scan_box<false>(b[c], re, tnear_e, INFINITY));
            if (S) ms = shift_in_sign(ms, scan_box<true>(b[c], rs, tnear_s, tfar_s));
        }
    }
    for (; k < n; k++) {
        float b[6];
#pragma unroll
        for (int j = 0; j < 6; j++) b[j] = sx.boxes[k * 8 + j];
        if (E) me = shift_in_sign(me, scan_box<false>(b, re, tnear_e, INFINITY));
        if (S) ms = shift_in_sign(ms, scan_box<true>(b, rs, tnear_s, tfar_s));
    }
}
__device__ __forceinline__ void scan_leaf_boxes2(const ScanCtx &sx, f3 org, f3 dir_e, float tnear_e, f3 dir_s, float tnear_s, float tfar_s, bool any_e, bool any_s,
                                                 uint32_t &me, uint32_t &ms) {
    const ScanRay re = scan_ray(org, dir_e), rs = scan_ray(org, dir_s);
    tnear_e = fmaxf(tnear_e, 0.0f); tnear_s = fmaxf(tnear_s, 0.0f);
    me = 0u; ms = 0u;
    if (any_e && any_s) scan_leaf_boxes<true, true>(sx, re, tnear_e, rs, tnear_s, tfar_s, me, ms);
    else if (any_e) scan_leaf_boxes<true, false>(sx, re, tnear_e, rs, tnear_s, tfar_s, me, ms);
    else if (any_s) scan_leaf_boxes<false, true>(sx, re, tnear_e, rs, tnear_s, tfar_s, me, ms);
}

// Closest hit of every lane's extension ray (has_e) and occlusion of its shadow ray (has_s), wave-synchronous: all 64 lanes
// call it together.  Results: gprim (-1: miss), t, u, v exactly as k_extend reports them; occluded.
template <bool SPHERES>
__device__ __forceinline__ void scan_trace(const ScanCtx &sx, bool has_e, bool has_s, f3 org, f3 dir_e, float tnear_e, f3 dir_s, float tnear_s, float tfar_s,
                                           float &ht, float &hu, float &hv, int &gprim_out, bool &occluded) {
    // No floating-point contraction in here: u = U * (1 / S) must reach the shading code as the rounded product k_extend stores in
    // the hit record, not as a multiply the compiler may fuse into the consumer's `1 - u - v` (a 1-ulp difference in 1e-4 of the samples).
#pragma clang fp contract(off)
    const uint32_t lane = lane_id();
    uint32_t me = 0, ms = 0;
    scan_leaf_boxes2(sx, org, dir_e, tnear_e, dir_s, tnear_s, tfar_s, __ballot(has_e) != 0ull, __ballot(has_s) != 0ull, me, ms);
    me = has_e ? me : 0u; ms = has_s ? ms : 0u;
    v4f r0, r1, r2;
    r0.x = org.x; r0.y = org.y; r0.z = org.z; r0.w = tfar_s;
    r1.x = dir_e.x; r1.y = dir_e.y; r1.z = dir_e.z; r1.w = tnear_e;
    r2.x = dir_s.x; r2.y = dir_s.y; r2.z = dir_s.z; r2.w = tnear_s;
    sx.rays[lane * 3] = r0; sx.rays[lane * 3 + 1] = r1; sx.rays[lane * 3 + 2] = r2;
    sx.keys[lane] = ~0ull; sx.occl[lane] = 0u;
    for (;;) {
        // ---- pool the candidates of all lanes: round j takes every lane's j-th candidate (shadow ray first)
        uint32_t n_items = 0;
        for (;;) {
            const bool has = (ms | me) != 0u;
            const unsigned long long b = __ballot(has);
            if (b == 0ull || n_items + 64u > kItemCap) break;
            if (has) {
                uint32_t leaf, kind;
                if (ms) { leaf = (uint32_t)(sx.n_used - 1) - (uint32_t)__builtin_ctz(ms); ms &= ms - 1u; kind = 1u; }
                else { leaf = (uint32_t)(sx.n_used - 1) - (uint32_t)__builtin_ctz(me); me &= me - 1u; kind = 0u; }
                const uint32_t idx = n_items + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                sx.items[idx] = (uint16_t)(lane | (leaf << 6) | (kind << 11));
            }
            n_items += (uint32_t)__popcll(b);
        }
        if (n_items == 0u) break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        // ---- test them, 64 at a time, whichever lane is free
        for (uint32_t r = 0; r < n_items; r += 64u) {
            if (r + lane < n_items) {
                const uint32_t it = sx.items[r + lane];
                const uint32_t src = it & 63u, leaf = (it >> 6) & 31u, kind = it >> 11;
                const v4f o4 = sx.rays[src * 3], d4 = sx.rays[src * 3 + 1 + kind];
                RayF ray;
                ray.ox = o4.x; ray.oy = o4.y; ray.oz = o4.z; ray.dx = d4.x; ray.dy = d4.y; ray.dz = d4.z;
                ray.tnear = d4.w; ray.tfar = kind ? o4.w : INFINITY;
                const int first = sx.leaf_tab[2 * leaf], cnt = sx.leaf_tab[2 * leaf + 1];
                for (int p = 0; p < cnt; p++) {
                    const int pi = first + p, S = sx.prim_stride;
                    const v4f p0 = sx.lprims[pi], p1 = sx.lprims[S + pi], p2 = sx.lprims[2 * S + pi];
                    const int gprim = __float_as_int(p0.w);
                    bool hit; float t;
                    if (!SPHERES || __float_as_int(p1.w) == 0) {
                        const float v0[3] = {p0.x, p0.y, p0.z}, v1[3] = {p1.x, p1.y, p1.z}, v2[3] = {p2.x, p2.y, p2.z};
                        float U, V, Ssum;
                        hit = tri_test_raw(ray, ray.tfar, v0, v1, v2, t, U, V, Ssum);
                    } else {
                        double td = 0.0;
                        hit = sphere_test(ray, sx.spheres[__float_as_int(p2.w)], td);
                        t = (float)td;
                    }
                    if (hit) {
                        if (kind) sx.occl[src] = 1u;
                        else (void)__hip_atomic_fetch_min(&sx.keys[src], ((unsigned long long)f2u(t) << 32) | (unsigned long long)(((uint32_t)gprim << 16) | (uint32_t)pi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (__ballot((ms | me) != 0u) == 0ull) break;
    }
    // ---- every lane picks up its own results; the winner's barycentrics are recomputed (one test per ray instead of carrying
    // U, V, S through the pool)
    const unsigned long long key = sx.keys[lane];
    const uint32_t low = (uint32_t)key;
    occluded = sx.occl[lane] != 0u;
    gprim_out = -1; ht = 0.0f; hu = 0.0f; hv = 0.0f;
    if (has_e && low != 0xffffffffu) {
        const int pi = (int)(low & 0xffffu), S = sx.prim_stride;
        gprim_out = (int)(low >> 16);
        ht = u2f((uint32_t)(key >> 32));
        const v4f p0 = sx.lprims[pi], p1 = sx.lprims[S + pi], p2 = sx.lprims[2 * S + pi];
        if (!SPHERES || __float_as_int(p1.w) == 0) {
            RayF ray;
            ray.ox = org.x; ray.oy = org.y; ray.oz = org.z; ray.dx = dir_e.x; ray.dy = dir_e.y; ray.dz = dir_e.z; ray.tnear = tnear_e; ray.tfar = INFINITY;
            const float v0[3] = {p0.x, p0.y, p0.z}, v1[3] = {p1.x, p1.y, p1.z}, v2[3] = {p2.x, p2.y, p2.z};
            float t, U, V, Ssum;
            (void)tri_test_raw(ray, INFINITY, v0, v1, v2, t, U, V, Ssum);
            const float rS = div_ieee(1.0f, Ssum);   // (trav_finish of k_extend)
            hu = U * rS; hv = V * rS;
        }
    }
}

// a + b of two values that are each already rounded: the wavefront kernels add the next-event contribution after it has been through
// the queue, so the product that formed it must not be fused into this addition
__device__ __forceinline__ f3 add_rounded(f3 a, f3 b) {
#pragma clang fp contract(off)
    return mk3(a.x + b.x, a.y + b.y, a.z + b.z);
}

// The wavefront kernels hand a path from one step to the next through the queue, so every value shade_path receives is a rounded
// float the compiler knows nothing about.  Here the same values stay in registers across the loop; without this fence the optimiser
// may fuse the multiply that produced one of them into an addition that consumes it in the next step (a 1-ulp difference in one
// sample of 10^4).  An empty asm per field makes each an opaque register value: no instruction is emitted.
__device__ __forceinline__ void opaque(float &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void opaque(f3 &v) { opaque(v.x); opaque(v.y); opaque(v.z); }
__device__ __forceinline__ void opaque_state(PathState &ps) {
    opaque(ps.org); opaque(ps.dir); opaque(ps.ht); opaque(ps.hu); opaque(ps.hv); opaque(ps.sdir); opaque(ps.stfar);
    opaque(ps.W); opaque(ps.rr); opaque(ps.p2); opaque(ps.rad); opaque(ps.nee); opaque(ps.eta_scale); opaque(ps.spread);
}

// LDS image of the scan area at byte offset `at` (16-byte aligned): [leaf table][primitives, transposed][4 x wave scratch]
__device__ __forceinline__ ScanCtx stage_scan(const DScene &sc, uint32_t at) {
    ScanCtx sx;
    char *base = (char *)lj_smem + at;
    LJ_LDS int *lt = (LJ_LDS int *)base;
    for (int i = threadIdx.x; i < sc.n_scan_leaves; i += kBlock) { lt[2 * i] = sc.scan_leaves[i].first; lt[2 * i + 1] = sc.scan_leaves[i].count; }
    const uint32_t lt_bytes = ((uint32_t)sc.n_scan_leaves * 8u + 15u) & ~15u;
    LJ_LDS v4f *lp = (LJ_LDS v4f *)(base + lt_bytes);
    const v4f *src = reinterpret_cast<const v4f *>(sc.leaf_prims);
    for (int i = threadIdx.x; i < sc.n_prims * 3; i += kBlock) lp[(i % 3) * sc.n_prims + (i / 3)] = src[i];
    char *wave = base + lt_bytes + (uint32_t)sc.n_prims * 48u + (threadIdx.x >> 6) * kWaveScanBytes;
    sx.boxes = (const LJ_CONST float *)(uintptr_t)sc.scan_leaves; sx.n_used = sc.n_scan_used;
    sx.leaf_tab = lt; sx.lprims = lp; sx.prim_stride = sc.n_prims; sx.spheres = sc.spheres;
    sx.rays = (LJ_LDS v4f *)wave; sx.keys = (LJ_LDS unsigned long long *)(wave + 64 * 48); sx.occl = (LJ_LDS uint32_t *)(wave + 64 * 48 + 64 * 8);
    sx.items = (LJ_LDS uint16_t *)(wave + 64 * 48 + 64 * 8 + 64 * 4);
    return sx;
}
size_t scan_smem(int n_scan_leaves, int n_prims) { return (((size_t)n_scan_leaves * 8 + 15) & ~(size_t)15) + (size_t)n_prims * 48 + 4 * (size_t)kWaveScanBytes; }

// waves per SIMD the kernel is built for.  The Lambert-only instantiation needs 113 VGPRs unconstrained; built for five waves (96
// VGPRs, 7 of them spilled to scratch) it is 7 % faster than for four — the kernel waits on LDS round trips and dependent issue,
// which a fifth wave hides (measured on MI355X, cbox 256 spp: 3 waves 17.8 ms, 4: 16.1, 5: 14.9, 6: 15.7 with 43 spills).  The
// feature sets with textures, microfacet lobes or sphere lights need ~150 VGPRs and spill 25-70 registers already at four: three.
#ifndef LJ_MEGA_OCC
#define LJ_MEGA_OCC 5
#endif
template <class Ft> struct MegaOccupancy { static constexpr int waves = 3; };
template <> struct MegaOccupancy<FeatLambert> { static constexpr int waves = LJ_MEGA_OCC; };
#ifndef LJ_MEGA_OCC_PLASTIC
#define LJ_MEGA_OCC_PLASTIC 4
#endif
template <> struct MegaOccupancy<FeatPlastic> { static constexpr int waves = LJ_MEGA_OCC_PLASTIC; };

// stats: [0] bounce iterations, [1] closest-hit rays, [2] shadow rays, [3] samples finished, [4] path steps (shade_path calls)
template <class Ft, bool SPHERES>
__global__ void __launch_bounds__(kBlock, MegaOccupancy<Ft>::waves) k_mega(DScene sc, DPass pass, ShadeStage stg, uint32_t scan_at, uint32_t n_samples, uint32_t grab,
                                                              uint32_t *sample_counter, unsigned long long *stats) {
    stage_shade_tables<2>(sc, stg, 0u);
    const ScanCtx sx = stage_scan(sc, scan_at);
    __syncthreads();
    const uint32_t lane = lane_id();
    ShadeCounters cnt; cnt.bounces = cnt.closest = cnt.shadow = cnt.done = 0;
    uint32_t steps = 0;
    bool live = false, exhausted = false;
    uint32_t w_next = 0, w_end = 0;   // the wave's open range of camera samples (wave-uniform)
    PathState ps;
    ps.flags = 0u; ps.stfar = 0.0f; ps.sample = 0u;
    for (;;) {
        // ---- the step of a path that is under way: hit accounting, next-event estimation, BSDF sampling (path_tracing.h:58-322)
        opaque_state(ps);
        if (live) {
            steps++;
            if (!shade_path<Ft>(sc, pass, ps, cnt)) {
                float *o = pass.sample_rgb + 3ull * ps.sample;
                o[0] = ps.rad.x; o[1] = ps.rad.y; o[2] = ps.rad.z;
                cnt.done++; live = false;
            }
        }
        // ---- lanes without a path take the next camera samples (path_tracing.h:10-14)
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
                if (!live && rank < left) { generate_path(sc, pass, w_next + rank, ps); live = true; }
                w_next += n_dead < left ? n_dead : left;
            }
        }
        if (__ballot(live) == 0ull) { if (exhausted) break; else continue; }
        // ---- both rays of the vertex: the pending shadow ray [eps, (1 - eps) d] and the extension ray [eps, inf) (camera rays from 0)
        opaque_state(ps);
        const bool has_e = live && !(ps.flags & PF_NO_EXT), has_s = live && ps.stfar > 0.0f;
        float ht, hu, hv; int gprim; bool occluded;
        scan_trace<SPHERES>(sx, has_e, has_s, ps.org, ps.dir, (ps.flags & 0xffffu) == 2u ? 0.0f : sc.eps, ps.sdir, sc.eps, ps.stfar, ht, hu, hv, gprim, occluded);
        if (live) {
            if (has_s && !occluded) ps.rad = add_rounded(ps.rad, ps.nee);   // path_tracing.h:207 (k_shade adds it at the top of the next step)
            ps.ht = ht; ps.hu = hu; ps.hv = hv; ps.hcode = gprim + 1;
            if (ps.flags & PF_NO_EXT) {   // sample_bsdf failed (path_tracing.h:220-223): nothing left but the contribution just added
                float *o = pass.sample_rgb + 3ull * ps.sample;
                o[0] = ps.rad.x; o[1] = ps.rad.y; o[2] = ps.rad.z;
                cnt.done++; live = false;
            }
        }
    }
    const uint32_t b = wave_sum(cnt.bounces), cl = wave_sum(cnt.closest), sh = wave_sum(cnt.shadow), dn = wave_sum(cnt.done), sp = wave_sum(steps);
    if (lane == 0u) {
        atomicAdd(&stats[0], (unsigned long long)b); atomicAdd(&stats[1], (unsigned long long)cl); atomicAdd(&stats[2], (unsigned long long)sh);
        atomicAdd(&stats[3], (unsigned long long)dn); atomicAdd(&stats[4], (unsigned long long)sp);
    }
}

// batched intersect() / occluded() through the scan (the parity tests hold it bit for bit to the oracle like the BVH traversal)
struct RayIO { float org[3]; float tnear; float dir[3]; float tfar; };
struct HitIO { float t, u, v; int32_t shape_id, prim_id; };
__global__ void __launch_bounds__(kBlock) k_trace_rays_scan(DScene sc, const RayIO *rays, long long n, HitIO *hits, unsigned char *occ) {
    const ScanCtx sx = stage_scan(sc, 0u);
    __syncthreads();
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i0 = (long long)blockIdx.x * kBlock; i0 < n; i0 += stride) {   // wave-complete iterations
        const long long i = i0 + threadIdx.x;
        const bool act = i < n;
        f3 org = mk3(0, 0, 0), dir = mk3(0, 0, 1); float tn = 0.0f, tf = 0.0f;
        if (act) { org = ld3(rays[i].org); dir = ld3(rays[i].dir); tn = rays[i].tnear; tf = rays[i].tfar; }
        float ht, hu, hv; int gprim; bool occluded;
        if (occ) {   // any hit in (tnear, tfar]: the shadow-ray slot
            scan_trace<true>(sx, false, act, org, dir, 0.0f, dir, tn, tf, ht, hu, hv, gprim, occluded);
            if (act) occ[i] = occluded ? 1 : 0;
        } else {
            // closest hit in (tnear, tfar]: the extension-ray slot has tfar = inf, so the far end is applied to the result
            scan_trace<true>(sx, act, false, org, dir, tn, dir, 0.0f, 0.0f, ht, hu, hv, gprim, occluded);
            if (act) {
                HitIO o; o.t = 0; o.u = 0; o.v = 0; o.shape_id = -1; o.prim_id = -1;
                if (gprim >= 0 && ht <= tf) { const DPrimShade &ps = sc.prims[gprim]; o.t = ht; o.u = hu; o.v = hv; o.shape_id = ps.shape_id; o.prim_id = ps.prim_id; }
                hits[i] = o;
            }
        }
    }
}

// ---------------------------------------------------------------- launchers

// LDS a k_mega workgroup needs (0: the scene cannot run as a mega launch)
size_t mega_smem(const DScene &sc, const ShadeConfig &scfg) {
    if (sc.n_scan_leaves <= 0 || !scfg.stage_prims || scfg.smem == 0) return 0;
    const size_t at = (scfg.smem + 15) & ~(size_t)15, total = at + scan_smem(sc.n_scan_leaves, sc.n_prims);
    return total <= 64 * 1024 ? total : 0;
}

// workgroups per CU a mega launch keeps resident (the grid is persistent: n_cus * this)
int mega_blocks_per_cu(const ShadeConfig &scfg) {
    int waves = 3;
    with_shade_variant(scfg.variant, [&](auto ft) { waves = MegaOccupancy<decltype(ft)>::waves; });
    return waves;
}

void launch_mega(const DScene &sc, const DPass &pass, const ShadeConfig &scfg, bool spheres, uint32_t n_samples, uint32_t grab, uint32_t *sample_counter,
                 unsigned long long *stats, int grid, hipStream_t s) {
    const ShadeStage st = make_shade_stage(scfg);
    const uint32_t at = (uint32_t)((scfg.smem + 15) & ~(size_t)15);
    const size_t smem = mega_smem(sc, scfg);
    with_shade_variant(scfg.variant, [&](auto ft) {
        using Ft = decltype(ft);
        if (spheres) hipLaunchKernelGGL((k_mega<Ft, true>), dim3(grid), dim3(kBlock), smem, s, sc, pass, st, at, n_samples, grab, sample_counter, stats);
        else hipLaunchKernelGGL((k_mega<Ft, false>), dim3(grid), dim3(kBlock), smem, s, sc, pass, st, at, n_samples, grab, sample_counter, stats);
    });
}

void launch_trace_rays_scan(const DScene &sc, const void *rays, long long n, void *hits, unsigned char *occ, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_trace_rays_scan, dim3(grid), dim3(kBlock), scan_smem(sc.n_scan_leaves, sc.n_prims), s, sc, (const RayIO *)rays, n, (HitIO *)hits, occ);
}

} // namespace ljd
