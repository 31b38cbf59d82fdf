// BVH4 traversal steps shared by the kernels that walk the flattened tree (kernels.hip: k_extend, k_tail, k_trace_rays, k_aux;
// volpath.hip: the volumetric path tracer): the LDS image of a workgroup (per-lane stacks, the top of the tree, small scenes' primitives),
// one inner-node step, one leaf step, the pooled leaf phase.  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include "dpool.h"

namespace ljd {

// ---------------------------------------------------------------- extend
// LDS image (dynamic): [ per-lane stacks: cap x 256 ints ][ first n_lnodes BVH4 nodes ][ first n_lprims leaf prims ]
// Nodes are stored breadth-first, so a prefix of the node array is the top of the tree.
struct TreeView {
    const char *gnodes; const v4f *gprims; const DSphere *spheres;
    const LJ_LDS char *lnodes; const LJ_LDS v4f *lprims;
    LJ_LDS int *stack;   // this lane's column; level l lives at stack[l * kBlock]
    int *spill;          // this lane's column of the overflow stack in global memory (levels >= cap), stride spill_stride
    uint32_t spill_stride;
    int n_lnodes, n_lprims, prim_stride, cap;
    uint32_t qstride;    // bytes between two quarters of one node in the LDS image (= staged nodes * 16)
};


__device__ __forceinline__ TreeView stage_tree(const DScene &sc, int stack, int lds_nodes, int lds_prims, int *spill, uint32_t spill_stride, uint32_t lane_global) {
    TreeView tv;
    LJ_LDS v4f *base = (LJ_LDS v4f *)lj_smem;
    tv.stack = (LJ_LDS int *)base + threadIdx.x;
    tv.cap = stack;
    tv.spill = spill + lane_global; tv.spill_stride = spill_stride;
    LJ_LDS v4f *ln = base + (stack * kBlock) / 4;
    LJ_LDS v4f *lp = ln + lds_nodes * 7;
    tv.n_lnodes = sc.n_nodes < lds_nodes ? sc.n_nodes : lds_nodes;
    tv.n_lprims = sc.n_prims < lds_prims ? sc.n_prims : lds_prims;
    tv.qstride = (uint32_t)lds_nodes * 16u; tv.prim_stride = lds_prims;
    // The LDS image is transposed: quarter k of node i sits at ln[k * lds_nodes + i].  Lanes that fetch different
    // nodes then hit different 16-byte bank slots (a 128-byte stride would put every node on the same four).
    const v4f *src = reinterpret_cast<const v4f *>(sc.nodes);
    for (int i = threadIdx.x; i < tv.n_lnodes * 7; i += kBlock) { const int node = i / 7, k = i - node * 7; ln[k * lds_nodes + node] = src[node * 8 + k]; }
    src = reinterpret_cast<const v4f *>(sc.leaf_prims);
    for (int i = threadIdx.x; i < tv.n_lprims * 3; i += kBlock) lp[(i % 3) * lds_prims + (i / 3)] = src[i];
    __syncthreads();
    tv.gnodes = reinterpret_cast<const char *>(sc.nodes); tv.gprims = reinterpret_cast<const v4f *>(sc.leaf_prims);
    tv.spheres = sc.spheres; tv.lnodes = (const LJ_LDS char *)ln; tv.lprims = lp;
    return tv;
}

constexpr int kDone = 0x7fffffff;  // "no more work for this ray" marker in `cur`

struct LaneTrav {
    RayF ray;
    // slab constants: t = fma(plane, i, -oi) with i = 1 / d, oi = o * i
    float ix, iy, iz, oix, oiy, oiz;
    HitRec best;     // for a triangle hit u, v hold the unnormalised barycentrics U, V until trav_finish divides by best_S
    float best_S;
    int cur, sp;
    int held;        // a leaf this ray has reached but not tested yet (0: none) — see trav_hold
    uint32_t nqx, nqy, nqz;  // quarter index (0..5) holding the NEAR plane of each axis for this ray's direction signs
};

__device__ __forceinline__ void trav_begin(LaneTrav &L, float tnear, float tfar) {
    L.ray.tnear = tnear; L.ray.tfar = tfar;
    // v_rcp_f32 (1 ulp) is enough here: the slabs only steer the traversal, and the boxes carry a 1e-5 pad plus a 4-ulp
    // widening of the exit distance; hits are decided by the primitive tests alone
    L.ix = __builtin_amdgcn_rcpf(L.ray.dx); L.iy = __builtin_amdgcn_rcpf(L.ray.dy); L.iz = __builtin_amdgcn_rcpf(L.ray.dz);
    // One fma per plane instead of subtract + multiply.  (plane - o) * i is exact where plane ~ o and the fma is not, but
    // its error there, ulp(o * i), is a hundredth of what the builder's 1e-5 box padding amounts to in t; away from that
    // both forms carry the same rounding of o.  A direction component of 0 gives i = inf and o * i = inf or nan: the
    // planes of that axis then all read nan, which fmax / fmin ignore — the axis drops out of the test (conservative).
    L.oix = L.ray.ox * L.ix; L.oiy = L.ray.oy * L.iy; L.oiz = L.ray.oz * L.iz;
    L.nqx = L.ix < 0.0f ? 3u : 0u; L.nqy = L.iy < 0.0f ? 4u : 1u; L.nqz = L.iz < 0.0f ? 5u : 2u;
    L.best.t = tfar; L.best.u = 0.0f; L.best.v = 0.0f; L.best.gprim = -1; L.best_S = 1.0f;
    L.cur = 0; L.sp = 0; L.held = 0;
}
__device__ __forceinline__ void trav_push(const TreeView &tv, LaneTrav &L, int v) {
    if (L.sp < tv.cap) tv.stack[L.sp * kBlock] = v;
    else tv.spill[(uint32_t)(L.sp - tv.cap) * tv.spill_stride] = v;
    L.sp++;
}
template <bool RESIDENT = false>
__device__ __forceinline__ int trav_pop(const TreeView &tv, LaneTrav &L) {
    // every level (of the lanes that pop here) is in LDS: one read, no branch (and no global load for the scheduler to wait on)
    if (RESIDENT || __ballot(L.sp > tv.cap) == 0ull) {
        const int sp = L.sp > 0 ? L.sp - 1 : 0;
        const int v = tv.stack[sp * kBlock];
        const int r = L.sp > 0 ? v : kDone;
        L.sp = sp;
        return r;
    }
    if (L.sp == 0) return kDone;
    L.sp--;
    if (L.sp < tv.cap) return tv.stack[L.sp * kBlock];
    return tv.spill[(uint32_t)(L.sp - tv.cap) * tv.spill_stride];
}

// A ray that reaches a leaf sets it aside and goes on with the next entry of its stack; it only has to wait for the wave's leaf phase when
// it reaches a second one.  The lanes of a wave then spend more node steps together before the leaf phase (which finds fuller rounds), at
// the price of the node steps a hit in the held leaf would have culled.  The closest hit is the minimum of (t, primitive id) over
// everything the ray tests, so the order of the tests cannot change it.
#ifndef LJ_EXT_HOLD
#define LJ_EXT_HOLD 1
#endif
#ifndef LJ_EXT_HOLD_SHADOW
#define LJ_EXT_HOLD_SHADOW 1   // any-hit rays hold a leaf too (0: they wait at their first leaf — a hit there ends them)
#endif
template <bool RESIDENT>
__device__ __forceinline__ void trav_hold(const TreeView &tv, LaneTrav &L) {
    L.held = L.cur;
    L.cur = trav_pop<RESIDENT>(tv, L);
}

// the one barycentric division of a closest-hit query (dtrace.h tri_test: u = U * (1 / S))
__device__ __forceinline__ void trav_finish(LaneTrav &L) {
    const float rS = div_ieee(1.0f, L.best_S);
    L.best.u = L.best.u * rS; L.best.v = L.best.v * rS;
}

#ifndef LJ_EXT_CSW_MINMAX
#define LJ_EXT_CSW_MINMAX 0   // 1: the distances of a compare-exchange are taken as min / max (nothing waits on the comparison's mask)
#endif
#ifndef LJ_EXT_SORT
#define LJ_EXT_SORT 5         // compare-exchanges of a node step: 5 = the four children fully ordered; 4 = nearest and farthest in place; 3 = nearest only
#endif
__device__ __forceinline__ void csw(float &ta, int &ca, float &tb, int &cb) {  // compare-exchange: nearer entry first
    const bool sw = tb < ta;
#if LJ_EXT_CSW_MINMAX
    const float t0 = fminf(ta, tb), t1 = fmaxf(ta, tb);   // (entry distances are numbers: same values as the selects)
#else
    const float t0 = sw ? tb : ta, t1 = sw ? ta : tb;
#endif
    const int c0 = sw ? cb : ca, c1 = sw ? ca : cb;
    ta = t0; tb = t1; ca = c0; cb = c1;
}

// one inner-node step: slab-test the four children, continue with the nearest one that is hit, push the others far-first.
// The stack part of the step is free of branches whenever the three pushes of every lane stay inside the LDS levels (always, when
// the scene is RESIDENT — the whole tree and every stack level in LDS; else a wave-uniform test, which only a ray deeper than
// `cap - 3` entries fails): the pushes store unconditionally and advance `sp` only for a hit, and the pop candidate is fetched with the node.
#ifndef LJ_EXT_LDS_NODES
#define LJ_EXT_LDS_NODES 1   // 0: a tree that is not fully LDS-resident is read through L1 / L2 only (no LDS copy of its top)
#endif
template <bool RESIDENT>
__device__ __forceinline__ void trav_node_step(const TreeView &tv, LaneTrav &L) {
    v4f nx, ny, nz, fx, fy, fz, ch;
    const int i = L.cur;
    int popped = kDone;
    const bool fast = RESIDENT || __ballot(L.sp + 3 > tv.cap) == 0ull;
    if (RESIDENT || (LJ_EXT_LDS_NODES && i < tv.n_lnodes)) {
        const LJ_LDS char *b = tv.lnodes + (uint32_t)i * 16u;
        const uint32_t S = tv.qstride;
        nx = *(const LJ_LDS v4f *)(b + L.nqx * S); fx = *(const LJ_LDS v4f *)(b + (3u - L.nqx) * S);
        ny = *(const LJ_LDS v4f *)(b + L.nqy * S); fy = *(const LJ_LDS v4f *)(b + (5u - L.nqy) * S);
        nz = *(const LJ_LDS v4f *)(b + L.nqz * S); fz = *(const LJ_LDS v4f *)(b + (7u - L.nqz) * S);
        ch = *(const LJ_LDS v4f *)(b + 6u * S);
    } else {
        const char *g = tv.gnodes;
        const uint32_t o = (uint32_t)i * 128u;
        nx = *(const v4f *)(g + (o + L.nqx * 16u)); fx = *(const v4f *)(g + (o + (3u - L.nqx) * 16u));
        ny = *(const v4f *)(g + (o + L.nqy * 16u)); fy = *(const v4f *)(g + (o + (5u - L.nqy) * 16u));
        nz = *(const v4f *)(g + (o + L.nqz * 16u)); fz = *(const v4f *)(g + (o + (7u - L.nqz) * 16u));
        ch = *(const v4f *)(g + (o + 96u));
    }
    if (fast) popped = tv.stack[(L.sp > 0 ? L.sp - 1 : 0) * kBlock];
    const float inf = __builtin_inff();

    float t0[4]; int c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float te = fmaxf(fmaxf(__builtin_fmaf(nx[k], L.ix, -L.oix), __builtin_fmaf(ny[k], L.iy, -L.oiy)), fmaxf(__builtin_fmaf(nz[k], L.iz, -L.oiz), L.ray.tnear));
        const float tx = fminf(fminf(__builtin_fmaf(fx[k], L.ix, -L.oix), __builtin_fmaf(fy[k], L.iy, -L.oiy)), fminf(__builtin_fmaf(fz[k], L.iz, -L.oiz), L.best.t));
        t0[k] = (te <= tx * 1.0000005f) ? te : inf;
        c[k] = __float_as_int(ch[k]);
    }
    // (an entry distance is finite for a hit: it is bounded by best.t or by the finite far planes)
    csw(t0[0], c[0], t0[1], c[1]); csw(t0[2], c[2], t0[3], c[3]);
    csw(t0[0], c[0], t0[2], c[2]);
#if LJ_EXT_SORT >= 4
    csw(t0[1], c[1], t0[3], c[3]);
#endif
#if LJ_EXT_SORT >= 5
    csw(t0[1], c[1], t0[2], c[2]);
#endif
    if (fast) {
        const int sp0 = L.sp;
        tv.stack[L.sp * kBlock] = c[3]; L.sp += (t0[3] < inf) ? 1 : 0;   // misses sort last: a slot written for a miss is
        tv.stack[L.sp * kBlock] = c[2]; L.sp += (t0[2] < inf) ? 1 : 0;   // overwritten by the next store or never read
        tv.stack[L.sp * kBlock] = c[1]; L.sp += (t0[1] < inf) ? 1 : 0;
        const bool any = t0[0] < inf;
        L.cur = any ? c[0] : (sp0 > 0 ? popped : kDone);
        L.sp = any ? L.sp : (sp0 > 0 ? sp0 - 1 : 0);
    } else {
        if (t0[3] < inf) trav_push(tv, L, c[3]);
        if (t0[2] < inf) trav_push(tv, L, c[2]);
        if (t0[1] < inf) trav_push(tv, L, c[1]);
        L.cur = (t0[0] < inf) ? c[0] : trav_pop(tv, L);
    }
}

// one leaf: up to 8 primitives
// SPHERES: the scene holds sphere shapes (their test is the reference's double-precision callback; a scene without
// spheres should not even carry its set-up code).
template <bool RESIDENT, bool SPHERES>
__device__ __forceinline__ void trav_leaf_step(const TreeView &tv, LaneTrav &L, const bool ANY_HIT) {
    const int code = ~L.cur;
    const int first = code >> 3, count = (code & 7) + 1;
    bool stop = false;
    for (int k = 0; k < count && !stop; k++) {
        const int pi = first + k;
        v4f p0, p1, p2;
        if (RESIDENT || pi < tv.n_lprims) { const int S = tv.prim_stride; p0 = tv.lprims[pi]; p1 = tv.lprims[S + pi]; p2 = tv.lprims[2 * S + pi]; }
        else { p0 = tv.gprims[3 * pi]; p1 = tv.gprims[3 * pi + 1]; p2 = tv.gprims[3 * pi + 2]; }
        const int gprim = __float_as_int(p0.w), kind = __float_as_int(p1.w);
        if (!SPHERES || kind == 0) {
            const float v0[3] = {p0.x, p0.y, p0.z}, v1[3] = {p1.x, p1.y, p1.z}, v2[3] = {p2.x, p2.y, p2.z};
            float t = 0.0f, U = 0.0f, V = 0.0f, S = 1.0f;
            const bool hit = tri_test_raw(L.ray, L.best.t, v0, v1, v2, t, U, V, S);
            // selects, not branches: an any-hit ray only needs `gprim` (and stops), a closest-hit ray takes the nearer of
            // (t, gprim); what the other fields of an any-hit ray hold no longer matters
            const bool take = hit & (ANY_HIT | (t < L.best.t) | ((t == L.best.t) & ((L.best.gprim < 0) | (gprim < L.best.gprim))));
            L.best.t = take ? t : L.best.t; L.best.u = take ? U : L.best.u; L.best.v = take ? V : L.best.v;
            L.best_S = take ? S : L.best_S; L.best.gprim = take ? gprim : L.best.gprim;
            stop = hit & ANY_HIT;
        } else {
            double td;
            if (sphere_test(L.ray, tv.spheres[__float_as_int(p2.w)], td)) {
                const float tf = (float)td;
                // (selects here too: hipcc 7.2 mis-structurises the branch form `else if (a || (b && (c || d))) { five assignments }` —
                // lanes that take the tie arm kept their old u, v)
                const bool take = ANY_HIT | (tf < L.best.t) | ((tf == L.best.t) & ((L.best.gprim < 0) | (gprim < L.best.gprim)));
                L.best.t = take ? tf : L.best.t; L.best.u = take ? 0.0f : L.best.u; L.best.v = take ? 0.0f : L.best.v;
                L.best_S = take ? 1.0f : L.best_S; L.best.gprim = take ? gprim : L.best.gprim;
                stop = ANY_HIT;
            }
        }
    }
    L.cur = stop ? kDone : trav_pop<RESIDENT>(tv, L);
}

// Called by the whole wave.  `at_leaf`: this lane's L.cur is a leaf.  Returns the number of
// 64-pair rounds it ran (for the statistics); `n_pairs` the pairs.
template <bool RESIDENT, bool SPHERES>
__device__ __forceinline__ uint32_t trav_leaf_pool(const TreeView &tv, const LeafPool &lp, LaneTrav &L, const bool at_leaf, const bool any_hit, uint32_t &n_pairs) {
    const uint32_t lane = threadIdx.x & 63u;
    // this lane's leaves: the one it holds (A) and the one it sits on (B)
    const int codeA = ~L.held, codeB = ~L.cur;
    const int firstA = codeA >> 3, cntA = (at_leaf && L.held != 0) ? (codeA & 7) + 1 : 0;
    const int firstB = codeB >> 3, cntB = (at_leaf && L.cur < 0) ? (codeB & 7) + 1 : 0;
    const int count = cntA + cntB;
    uint32_t rounds = 0;
    n_pairs = 0;
    for (int j = 0;;) {
        // ---- list the pairs: pass j takes the j-th primitive of every leaf (neighbouring pairs then belong to different rays)
        uint32_t n_items = 0;
        for (;;) {
            const bool has = count > j;
            const unsigned long long b = __ballot(has);
            if (b == 0ull || n_items + 64u > kPoolCap) break;
            if (has) lp.items[n_items + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = lane | ((uint32_t)(j < cntA ? firstA + j : firstB + (j - cntA)) << 6);
            n_items += (uint32_t)__popcll(b); j++;
        }
        if (n_items == 0u) break;
        n_pairs += n_items;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        // ---- test them
        pool_test_items<RESIDENT, SPHERES>(tv, lp, L, n_items, rounds);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    // ---- every owner picks up its result
    if (at_leaf) {
        const bool stop = pool_collect(lp, L, any_hit);
        const bool on_leaf = L.cur < 0;
        L.held = 0;
        if (stop) L.cur = kDone;
        else if (on_leaf) L.cur = trav_pop<RESIDENT>(tv, L);
    }
    return rounds;
}

} // namespace ljd
