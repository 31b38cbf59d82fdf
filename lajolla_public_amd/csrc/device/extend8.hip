// gfx950 extend kernel for trees beyond the LDS image: closest hit of the extension ray + any hit of the pending shadow ray over the
// BVH8 with quantised child boxes (dtypes.h DNode8), and the batched intersect() / occluded() queries over the same tree.
// (intersection.cpp:7-85; what rtcIntersect1 / rtcOccluded1 do for the reference)
#include <hip/hip_runtime.h>
#include "dpool.h"

namespace ljd {

// ---------------------------------------------------------------- extend over the BVH8 (trees beyond the LDS image)
// The same wave-synchronous scheme as k_extend — persistent waves, dynamic refill, a node phase and a pooled leaf phase — over DNode8
// (dtypes.h): a step gathers 80 bytes (five dwordx4) instead of 112 and tests eight quantised children, a ray takes ~0.7x as many steps
// (sponza: 12.3 instead of 17.0 per extension ray), and the stack holds node GROUPS (child_base, remaining hit children in octant order |
// imask << 8) — at most one push per step, one 8-byte entry.  Leaf children that are hit are not tested at once: the lane notes
// (node, hit leaf slots) as a pending group and keeps descending; a lane with two pending groups waits for the wave's pooled leaf phase.
// LDS image (dynamic): [ per-lane group stacks: cap x 256 x 8 B ][ the first n_lnodes nodes, 80 B each ][ leaf pools ].
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2u mk2u(uint32_t x, uint32_t y) { v2u r; r.x = x; r.y = y; return r; }
struct Tree8View {
    const char *gnodes; const v4f *gprims; const DSphere *spheres;
    const LJ_LDS char *lnodes; uint32_t n_lnodes; uint32_t gstride;   // gstride: bytes between two nodes in global memory
    LJ_LDS v2u *stack;              // this lane's column; level l lives at stack[l * kBlock]
    v2u *spill; uint32_t spill_stride; int cap;
    const LJ_LDS v4f *lprims; int n_lprims, prim_stride;   // (no primitives are staged for a tree of this size; pool_test_items reads these)
};
__device__ __forceinline__ Tree8View stage_tree8(const DScene &sc, int stack, int lds_nodes, unsigned long long *spill, uint32_t spill_stride, uint32_t lane_global) {
    Tree8View tv;
    LJ_LDS v4f *base = (LJ_LDS v4f *)lj_smem;
    tv.stack = (LJ_LDS v2u *)base + threadIdx.x; tv.cap = stack;
    tv.spill = (v2u *)spill + lane_global; tv.spill_stride = spill_stride;
    LJ_LDS v4f *ln = base + (stack * kBlock * 8) / 16;
    tv.n_lnodes = (uint32_t)(sc.n_nodes8 < lds_nodes ? sc.n_nodes8 : lds_nodes);
    tv.gstride = (uint32_t)sc.node8_stride;
    const char *src = reinterpret_cast<const char *>(sc.nodes8);
    for (uint32_t i = threadIdx.x; i < tv.n_lnodes * 5u; i += kBlock) { const uint32_t node = i / 5u, k = i - node * 5u; ln[i] = *(const v4f *)(src + node * tv.gstride + k * 16u); }
    __syncthreads();
    tv.gnodes = reinterpret_cast<const char *>(sc.nodes8); tv.gprims = reinterpret_cast<const v4f *>(sc.leaf_prims); tv.spheres = sc.spheres;
    tv.lnodes = (const LJ_LDS char *)ln; tv.lprims = nullptr; tv.n_lprims = 0; tv.prim_stride = 0;
    return tv;
}

struct LaneTrav8 {
    RayF ray;
    Ray8 r8;            // 1 / d, o / d, direction octant
    HitRec best;        // u, v hold the unnormalised barycentrics until trav8_finish
    float best_S;
    uint32_t gbase, gbits;   // the open node group; (gbits & 0xff) == 0: none (then the stack is empty too)
    uint32_t pendA, pendB;   // pending leaf groups: node index | hit leaf slots << 24; 0: none (B only when A is set)
    int sp;
};
__device__ __forceinline__ void trav8_begin(LaneTrav8 &L, float tnear, float tfar) {
    L.ray.tnear = tnear; L.ray.tfar = tfar;
    L.r8 = ray8_setup(L.ray);
    L.best.t = tfar; L.best.u = 0.0f; L.best.v = 0.0f; L.best.gprim = -1; L.best_S = 1.0f;
    L.gbase = 0u; L.gbits = 1u; L.pendA = 0u; L.pendB = 0u; L.sp = 0;   // the root as a group of one (imask 0: the child picked is node gbase)
}
__device__ __forceinline__ void trav8_finish(LaneTrav8 &L) {
    const float rS = div_ieee(1.0f, L.best_S);
    L.best.u = L.best.u * rS; L.best.v = L.best.v * rS;
}
__device__ __forceinline__ bool trav8_descending(const LaneTrav8 &L) { return (L.gbits & 0xffu) != 0u && L.pendB == 0u; }
__device__ __forceinline__ bool trav8_finished(const LaneTrav8 &L) { return (L.gbits & 0xffu) == 0u && L.pendA == 0u; }

// one node step: take the next child of the open group, put the rest of the group on the stack, slab-test the child's eight children,
// open the group of its inner hits (or go back to the stack) and note its leaf hits.  Branch-free whenever every lane's push stays in the LDS levels.
__device__ __forceinline__ void trav8_node_step(const Tree8View &tv, LaneTrav8 &L) {
    const uint32_t k = (uint32_t)__builtin_ctz(L.gbits & 0xffu), s = k ^ L.r8.oct;
    const uint32_t rest = L.gbits & (L.gbits - 1u);
    const uint32_t node = L.gbase + (uint32_t)__builtin_popcount((rest >> 8) & ((1u << s) - 1u));
    const bool pushed = (rest & 0xffu) != 0u;
    const bool fast = __ballot(L.sp >= tv.cap) == 0ull;
    v2u top = mk2u(0u, 0u);
    const int sp0 = L.sp;
    if (fast) {
        top = tv.stack[(sp0 > 0 ? sp0 - 1 : 0) * kBlock];
        tv.stack[sp0 * kBlock] = mk2u(L.gbase, rest);
    } else if (pushed) {
        if (sp0 < tv.cap) tv.stack[sp0 * kBlock] = mk2u(L.gbase, rest);
        else tv.spill[(uint32_t)(sp0 - tv.cap) * tv.spill_stride] = mk2u(L.gbase, rest);
    } else if (sp0 > 0) {
        top = sp0 - 1 < tv.cap ? tv.stack[(sp0 - 1) * kBlock] : tv.spill[(uint32_t)(sp0 - 1 - tv.cap) * tv.spill_stride];
    }
    v4f q0, q1, q2, q3, q4;
    if (node < tv.n_lnodes) {
        const LJ_LDS v4f *b = (const LJ_LDS v4f *)(tv.lnodes + __umul24(node, 80u));
        q0 = b[0]; q1 = b[1]; q2 = b[2]; q3 = b[3]; q4 = b[4];
    } else {
        const v4f *g = (const v4f *)(tv.gnodes + __umul24(node, tv.gstride));
        q0 = g[0]; q1 = g[1]; q2 = g[2]; q3 = g[3]; q4 = g[4];
    }
    uint32_t w[20];
    w[0] = f2u(q0.x); w[1] = f2u(q0.y); w[2] = f2u(q0.z); w[3] = f2u(q0.w); w[4] = f2u(q1.x); w[5] = f2u(q1.y); w[6] = f2u(q1.z); w[7] = f2u(q1.w);
    w[8] = f2u(q2.x); w[9] = f2u(q2.y); w[10] = f2u(q2.z); w[11] = f2u(q2.w); w[12] = f2u(q3.x); w[13] = f2u(q3.y); w[14] = f2u(q3.z); w[15] = f2u(q3.w);
    w[16] = f2u(q4.x); w[17] = f2u(q4.y); w[18] = f2u(q4.z); w[19] = f2u(q4.w);
    const uint32_t hits = node8_hits(w, L.r8, L.ray.tnear, L.best.t);
    const uint32_t nmask = w[3] >> 24, inner = hits & nmask, leaf = hits & ~nmask;
    // leaf hits wait for the pooled phase (pendB is free here: a lane with two pending groups does not descend)
    const uint32_t pv = node | (leaf << 24);
    const bool isA = L.pendA == 0u, any_leaf = leaf != 0u;
    L.pendB = (any_leaf && !isA) ? pv : L.pendB;
    L.pendA = (any_leaf && isA) ? pv : L.pendA;
    // next group: the inner hits of this node, else the rest of the group just left, else the top of the stack
    const uint32_t opened = perm8(inner, L.r8.oct) | (nmask << 8);
    const bool any_inner = inner != 0u, from_stack = !any_inner && !pushed;
    L.gbase = any_inner ? w[4] : (pushed ? L.gbase : top.x);
    L.gbits = any_inner ? opened : (pushed ? rest : (sp0 > 0 ? top.y : 0u));
    L.sp = sp0 + ((any_inner && pushed) ? 1 : 0) - ((from_stack && sp0 > 0) ? 1 : 0);
}

// the primitives of a pending leaf group: base index and a mask of offsets (one bit per primitive; an empty slot that passed the widened
// box test of a degenerate node has no meta byte and contributes nothing)
__device__ __forceinline__ void trav8_expand(const Tree8View &tv, uint32_t pend, uint32_t &base, uint32_t &mask) {
    const uint32_t node = pend & 0xffffffu;
    uint32_t slots = pend >> 24;
    v4f q1;
    if (node < tv.n_lnodes) q1 = *(const LJ_LDS v4f *)(tv.lnodes + __umul24(node, 80u) + 16u);
    else q1 = *(const v4f *)(tv.gnodes + __umul24(node, tv.gstride) + 16u);
    base = f2u(q1.y);
    const unsigned long long meta = (unsigned long long)f2u(q1.z) | ((unsigned long long)f2u(q1.w) << 32);
    mask = 0u;
    while (slots) {
        const uint32_t i = (uint32_t)__builtin_ctz(slots); slots &= slots - 1u;
        const uint32_t m = (uint32_t)(meta >> (8u * i)) & 0xffu;
        mask |= (m & 0x80u) ? (((2u << ((m >> 5) & 3u)) - 1u) << (m & 31u)) : 0u;
    }
}

// the pooled leaf phase: called by the whole wave; `at_leaf`: this lane has pending leaf groups
template <bool SPHERES>
__device__ __forceinline__ uint32_t trav8_leaf_pool(const Tree8View &tv, const LeafPool &lp, LaneTrav8 &L, const bool at_leaf, const bool any_hit, uint32_t &n_pairs) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t baseA = 0u, mA = 0u, baseB = 0u, mB = 0u;
    if (at_leaf) {
        trav8_expand(tv, L.pendA, baseA, mA);
        if (L.pendB != 0u) trav8_expand(tv, L.pendB, baseB, mB);
    }
    uint32_t rounds = 0;
    n_pairs = 0;
    for (;;) {
        // ---- list the pairs: every pass takes the next primitive of every lane (neighbouring pairs then belong to different rays)
        uint32_t n_items = 0;
        for (;;) {
            const bool has = (mA | mB) != 0u;
            const unsigned long long b = __ballot(has);
            if (b == 0ull || n_items + 64u > kPoolCap) break;
            if (has) {
                const bool a = mA != 0u;
                const uint32_t m = a ? mA : mB, pi = (a ? baseA : baseB) + (uint32_t)__builtin_ctz(m);
                mA = a ? (mA & (mA - 1u)) : mA; mB = a ? mB : (mB & (mB - 1u));
                lp.items[n_items + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = lane | (pi << 6);
            }
            n_items += (uint32_t)__popcll(b);
        }
        if (n_items == 0u) break;
        n_pairs += n_items;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        pool_test_items<false, SPHERES>(tv, lp, L, n_items, rounds);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    if (at_leaf) {
        const bool stop = pool_collect(lp, L, any_hit);
        L.pendA = 0u; L.pendB = 0u;
        if (stop) { L.gbits = 0u; L.sp = 0; }
    }
    return rounds;
}

#ifndef LJ_EXT8_OCC
#define LJ_EXT8_OCC 5   // 91 - 95 VGPRs: five waves per SIMD, as many as five 30-KiB workgroups per CU bring
#endif
template <bool STATS, bool SPHERES>
__global__ void __launch_bounds__(kBlock, STATS ? 4 : LJ_EXT8_OCC) k_extend8(DScene sc, DQueue q, const DBlockState *blocks, uint32_t seg, uint32_t *work, const uint32_t *chunk_list, uint32_t parity, int stack, int lds_nodes, unsigned long long *spill, uint32_t refill_min, uint32_t min_descending, unsigned long long *stats, uint32_t pool_at) {
    unsigned long long st_outer = 0, st_busy = 0, st_nodes = 0, st_node_lanes = 0, st_leaf = 0, st_leaf_lanes = 0, st_refill = 0, st_rays = 0;
    const Tree8View tv = stage_tree8(sc, stack, lds_nodes, spill, gridDim.x * kBlock, blockIdx.x * kBlock + threadIdx.x);
    const LeafPool lp = leaf_pool_at(pool_at);
    // (work distribution, refill and ray set-up: exactly k_extend's — see the comments there)
    const uint32_t n_chunks = work[1 + parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) work[1 + (parity ^ 1u)] = 0u;
    uint32_t *chunk_counter = work;
    const bool leader = (threadIdx.x & 63u) == 0u;
    const uint32_t n_waves = gridDim.x * (kBlock / 64u);
    const bool draw = n_chunks > n_waves;
    uint32_t pre = blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6);
    uint32_t next = 0, end = 0;
    bool exhausted = false;
    bool busy = false; int phase = 0; uint32_t path = 0; uint32_t flags = 0; int vis = 0; int start = 0;
    float edx = 0, edy = 0, edz = 0;
    LaneTrav8 L; L.gbits = 0u; L.gbase = 0u; L.pendA = 0u; L.pendB = 0u; L.sp = 0;
    for (;;) {
        if (next == end && !exhausted) {
            const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)pre);
            if (c >= n_chunks) exhausted = true;
            else {
                const uint32_t slot0 = chunk_list[c] * kChunk, b = slot0 / seg, off = slot0 - b * seg, cnt = blocks[b].count;
                const uint32_t live = cnt > off ? (cnt - off < kChunk ? cnt - off : kChunk) : 0u;
                next = slot0; end = slot0 + live;
                if (draw) { if (leader) pre = atomicAdd(chunk_counter, 1u); }
                else pre = 0xffffffffu;
                if (live == 0u) continue;
            }
        }
        const unsigned long long idle = __ballot(!busy);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        const uint32_t left = end - next;
        if (left > 0 && (n_idle >= refill_min || n_idle == 64u)) {
            if (STATS) st_refill++;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const bool take = !busy && rank < left;
            if (take) {
                path = next + rank;
                const Rec4 ro = q.ro[path], rd = q.rd[path];
                L.ray.ox = ro.x; L.ray.oy = ro.y; L.ray.oz = ro.z;
                edx = rd.x; edy = rd.y; edz = rd.z; flags = f2u(rd.w);
                vis = 0; busy = true;
                if (ro.w > 0.0f) {
                    const Rec4 rs = q.rs[path];
                    L.ray.dx = rs.x; L.ray.dy = rs.y; L.ray.dz = rs.z; L.ray.tfar = ro.w;
                    start = 1;
                } else if (!(flags & PF_NO_EXT)) start = 2;
                else { q.rh[path] = mk4(0.0f, 0.0f, 0.0f, u2f(0u)); busy = false; }
            }
            next += n_idle < left ? n_idle : left;
        }
        if (start != 0) {
            const bool ext = start == 2;
            if (ext) { L.ray.dx = edx; L.ray.dy = edy; L.ray.dz = edz; }
            trav8_begin(L, (ext && (flags & 0xffffu) == 2u) ? 0.0f : sc.eps, ext ? INFINITY : L.ray.tfar);
            phase = ext ? 1 : 0; start = 0;
        }
        if (__ballot(busy) == 0ull) { if (exhausted && next == end) break; else continue; }
        if (STATS) { st_outer++; st_busy += __popcll(__ballot(busy)); }
        // ---- node phase: lanes with an open group and room for another pending leaf group step; once few of them are left and somebody
        // has leaves to test, the wave moves on
        for (;;) {
            const bool descending = busy && trav8_descending(L);
            const unsigned long long dm = __ballot(descending);
            if (dm == 0ull) break;
            if ((uint32_t)__popcll(dm) < min_descending && __ballot(busy && L.pendA != 0u) != 0ull) break;
            if (STATS) { st_nodes++; st_node_lanes += __popcll(dm); }
            if (descending) trav8_node_step(tv, L);
        }
        // ---- pooled leaf phase
        {
            uint32_t n_pairs;
            const uint32_t rounds = trav8_leaf_pool<SPHERES>(tv, lp, L, busy && L.pendA != 0u, phase == 0, n_pairs);
            if (STATS) { st_leaf += rounds; st_leaf_lanes += n_pairs; }
        }
        // ---- ray finished?
        const bool fin = busy && trav8_finished(L);
        if (STATS) st_rays += __popcll(__ballot(fin));
        if (fin) {
            if (phase == 0) {
                vis = (L.best.gprim < 0) ? HIT_VIS_BIT : 0;
                if (!(flags & PF_NO_EXT)) { start = 2; L.gbits = 1u; }   // (not "finished" any more: this block must not run twice)
                else { q.rh[path] = mk4(0.0f, 0.0f, 0.0f, u2f((uint32_t)vis)); busy = false; }
            } else {
                const uint32_t code = (uint32_t)vis | (uint32_t)(L.best.gprim + 1);
                const bool hit = L.best.gprim >= 0;
                trav8_finish(L);
                q.rh[path] = mk4(hit ? L.best.t : 0.0f, L.best.u, L.best.v, u2f(code));
                busy = false;
            }
        }
    }
    if (STATS && (threadIdx.x & 63) == 0) {
        atomicAdd(&stats[0], st_outer); atomicAdd(&stats[1], st_busy); atomicAdd(&stats[2], st_nodes); atomicAdd(&stats[3], st_node_lanes);
        atomicAdd(&stats[4], st_leaf); atomicAdd(&stats[5], st_leaf_lanes); atomicAdd(&stats[6], st_refill); atomicAdd(&stats[7], st_rays);
    }
}

// (the same queries over the BVH8, with k_extend8's two phases: what lj_intersect / lj_occluded run for a tree beyond the LDS image)
__global__ void __launch_bounds__(kBlock) k_trace_rays8(DScene sc, const RayIO *rays, long long n, HitIO *hits, unsigned char *occ, int stack, int lds_nodes, unsigned long long *spill, uint32_t pool_at) {
    const Tree8View tv = stage_tree8(sc, stack, lds_nodes, spill, gridDim.x * kBlock, blockIdx.x * kBlock + threadIdx.x);
    const LeafPool lp = leaf_pool_at(pool_at);
    for (long long i0 = (long long)blockIdx.x * kBlock; i0 < n; i0 += (long long)gridDim.x * kBlock) {
        const long long i = i0 + threadIdx.x;
        const bool act = i < n;
        LaneTrav8 L;
        L.ray.ox = 0.0f; L.ray.oy = 0.0f; L.ray.oz = 0.0f; L.ray.dx = 0.0f; L.ray.dy = 0.0f; L.ray.dz = 1.0f;
        if (act) {
            L.ray.ox = rays[i].org[0]; L.ray.oy = rays[i].org[1]; L.ray.oz = rays[i].org[2];
            L.ray.dx = rays[i].dir[0]; L.ray.dy = rays[i].dir[1]; L.ray.dz = rays[i].dir[2];
        }
        trav8_begin(L, act ? rays[i].tnear : 0.0f, act ? rays[i].tfar : 0.0f);
        if (!act) L.gbits = 0u;
        for (;;) {
            for (;;) {
                const bool descending = trav8_descending(L);
                if (__ballot(descending) == 0ull) break;
                if (descending) trav8_node_step(tv, L);
            }
            const bool at_leaf = L.pendA != 0u;
            if (__ballot(at_leaf) == 0ull) break;
            uint32_t n_pairs;
            (void)trav8_leaf_pool<true>(tv, lp, L, at_leaf, occ != nullptr, n_pairs);
        }
        trav8_finish(L);
        if (!act) continue;
        if (occ) occ[i] = L.best.gprim >= 0 ? 1 : 0;
        else {
            HitIO o; o.t = 0; o.u = 0; o.v = 0; o.shape_id = -1; o.prim_id = -1;
            if (L.best.gprim >= 0) {
                const DPrimShade &ps = sc.prims[L.best.gprim];
                o.t = L.best.t; o.u = L.best.u; o.v = L.best.v; o.shape_id = ps.shape_id; o.prim_id = ps.prim_id;
            }
            hits[i] = o;
        }
    }
}

// ---------------------------------------------------------------- launchers
size_t extend8_smem(const ExtendConfig &cfg) { return ((cfg.smem8 + 15) & ~(size_t)15) + (kBlock / 64) * kWavePoolBytes; }
void launch_extend8(const DScene &sc, const DQueue &q, const DBlockState *blocks, uint32_t grid, uint32_t seg, uint32_t *work, const uint32_t *chunk_list, uint32_t parity, const ExtendConfig &cfg, int *spill, unsigned long long *stats, hipStream_t s) {
    auto launch8 = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), extend8_smem(cfg), s, sc, q, blocks, seg, work, chunk_list, parity, cfg.stack8, cfg.lds_nodes8, (unsigned long long *)spill, cfg.refill_min, cfg.min_descending, stats,
                           (uint32_t)((cfg.smem8 + 15) & ~(size_t)15));
    };
    switch ((stats ? 2 : 0) | (cfg.spheres ? 1 : 0)) {
        case 0: launch8(k_extend8<false, false>); break;
        case 1: launch8(k_extend8<false, true>); break;
        case 2: launch8(k_extend8<true, false>); break;
        default: launch8(k_extend8<true, true>); break;
    }
}
void launch_trace_rays8(const DScene &sc, const void *rays, long long n, void *hits, unsigned char *occ, const ExtendConfig &cfg, int *spill, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_trace_rays8, dim3(grid), dim3(kBlock), extend8_smem(cfg), s, sc, (const RayIO *)rays, n, (HitIO *)hits, occ, cfg.stack8, cfg.lds_nodes8, (unsigned long long *)spill, (uint32_t)((cfg.smem8 + 15) & ~(size_t)15));
}

} // namespace ljd
