// gfx950 kernels of the wavefront integrator.  Written for wave64 / 256-thread workgroups / 160 KB LDS per CU.
//
//   k_extend    persistent   : closest hit of the extension ray + any hit of the pending shadow ray over the
//                              flattened BVH4.  Persistent waves draw 256-slot chunks of live paths from the list the
//                              preceding shade launch left; a lane whose rays are done pulls the next path instead of
//                              idling until the slowest lane of the wave finishes (while-while traversal with dynamic
//                              refill).  Top of the tree, leaf primitives of small scenes and the per-lane traversal
//                              stacks live in LDS.  (intersection.cpp:7-85)
//   k_shade     block/segment: hit accounting, NEE, BSDF sampling, Russian roulette (path_tracing.h:58-322).  Each
//                              workgroup owns one segment of the queue: survivors are compacted to the front of the
//                              segment in order (wave ballots + a 4-entry LDS scan), then the workgroup's next camera
//                              samples (path_tracing.h:10-14) are appended behind them and its live chunks listed.
//                              Instantiated per scene feature set (ShadeFeat).
//   k_tail      block/segment: the end of a render, fused: trace + shade + compact in a loop inside one launch
//   k_resolve   wave/pixel   : fixed-order sum of the per-sample radiance -> radiance / spp (render.cpp:94)
//   k_aux                    : the five auxiliary buffers (render.cpp:12-69)
//   k_volpath   lane/sample  : the volumetric path tracer (vol_path_tracing.h:503-869), one whole path per lane
//   k_trace_rays             : batched intersect()/occluded() for the parity tests
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <algorithm>
#include "dstage.h"
#include "dtrace.h"
#include "dconfig.h"
#include "dtrav.h"

namespace ljd {

// ---------------------------------------------------------------- queue <-> registers (16-byte records)
__device__ __forceinline__ void q_load_for_shade(const DQueue &q, uint32_t i, PathState &ps) {
    const Rec4 ro = q.ro[i], rd = q.rd[i], rh = q.rh[i], rw = q.rw[i], rl = q.rl[i], rn = q.rn[i], rg = q.rg[i];
    ps.org = mk3(ro.x, ro.y, ro.z); ps.dir = mk3(rd.x, rd.y, rd.z); ps.flags = f2u(rd.w);
    ps.ht = rh.x; ps.hu = rh.y; ps.hv = rh.z; ps.hcode = (int32_t)f2u(rh.w);
    ps.W = mk3(rw.x, rw.y, rw.z); ps.rr = rw.w;
    ps.rad = mk3(rl.x, rl.y, rl.z); ps.eta_scale = rl.w;
    ps.nee = mk3(rn.x, rn.y, rn.z); ps.spread = rn.w;
    ps.sample = f2u(rg.x); ps.rng = (uint64_t)f2u(rg.y) | ((uint64_t)f2u(rg.z) << 32); ps.p2 = rg.w;
    ps.sdir = mk3(0, 0, 0); ps.stfar = 0.0f;
}
__device__ __forceinline__ void q_store(const DQueue &q, uint32_t i, const PathState &ps) {
    q.ro[i] = mk4(ps.org.x, ps.org.y, ps.org.z, ps.stfar);
    q.rd[i] = mk4(ps.dir.x, ps.dir.y, ps.dir.z, u2f(ps.flags));
    q.rs[i] = mk4(ps.sdir.x, ps.sdir.y, ps.sdir.z, 0.0f);
    q.rw[i] = mk4(ps.W.x, ps.W.y, ps.W.z, ps.rr);
    q.rl[i] = mk4(ps.rad.x, ps.rad.y, ps.rad.z, ps.eta_scale);
    q.rn[i] = mk4(ps.nee.x, ps.nee.y, ps.nee.z, ps.spread);
    q.rg[i] = mk4(u2f(ps.sample), u2f((uint32_t)ps.rng), u2f((uint32_t)(ps.rng >> 32)), ps.p2);
}

// ---------------------------------------------------------------- extend (the traversal steps live in dtrav.h)
// STATS: developer instrumentation (LJ_EXTEND_STATS=1): wave-level step counts and the lanes active in them, summed into
// stats[0..7] = {outer iterations, sum of busy lanes, node steps, lanes in node steps, leaf prim rounds, lanes in them,
// refills, rays}.  The production instantiation carries none of it.
#ifndef LJ_EXT_RESIDENT_OCC
#define LJ_EXT_RESIDENT_OCC 4
#endif
template <bool STATS, bool RESIDENT, bool SPHERES>
#ifndef LJ_EXT_GENERAL_OCC
#define LJ_EXT_GENERAL_OCC 4
#endif
__global__ void __launch_bounds__(kBlock, (RESIDENT && !SPHERES && !STATS) ? LJ_EXT_RESIDENT_OCC : (STATS ? 4 : LJ_EXT_GENERAL_OCC)) k_extend(DScene sc, DQueue q, const DBlockState *blocks, uint32_t seg, uint32_t *work, const uint32_t *chunk_list, uint32_t parity, int stack, int lds_nodes, int lds_prims, int *spill, uint32_t refill_min, uint32_t min_descending, unsigned long long *stats, uint32_t pool_at) {
    unsigned long long st_outer = 0, st_busy = 0, st_nodes = 0, st_node_lanes = 0, st_leaf = 0, st_leaf_lanes = 0, st_refill = 0, st_rays = 0;
    const TreeView tv = stage_tree(sc, stack, lds_nodes, lds_prims, spill, gridDim.x * kBlock, blockIdx.x * kBlock + threadIdx.x);
#if LJ_EXT_POOL
    const LeafPool lp = leaf_pool_at(pool_at);
#endif
    // Persistent waves: the shade launch before this one listed the chunks (kChunk queue slots inside one of its
    // segments) that hold live paths.  Wave w starts with list entry w; if the list is longer than the grid has
    // waves, the rest is drawn from one grid-wide counter (which the shade launch preset to the number of waves), so
    // the launch stays balanced whatever the rays cost.  A launch over a short list issues no atomic at all — a
    // single address takes only ~90 atomics per microsecond, which would otherwise put a ~50 us floor under every
    // launch of the render's long tail.  work[0] = draw counter, work[1 + parity] = length of this step's list.
    const uint32_t n_chunks = work[1 + parity];
    if (blockIdx.x == 0 && threadIdx.x == 0) work[1 + (parity ^ 1u)] = 0u;   // the next shade launch appends to the other list
    uint32_t *chunk_counter = work;
    const bool leader = (threadIdx.x & 63u) == 0u;
    const uint32_t n_waves = gridDim.x * (kBlock / 64u);
    const bool draw = n_chunks > n_waves;   // uniform over the grid
    uint32_t pre = blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6);   // next list position (valid in the wave's first lane)
    uint32_t next = 0, end = 0;             // live slots of the open chunk (wave-uniform)
    bool exhausted = false;
    // per-lane state: phase 0 = shadow ray (any hit), phase 1 = extension ray (closest hit)
    // `start`: the ray this lane has to set up before it traverses again — 1 = its pending shadow ray, 2 = its extension
    // ray; written by the refill (a new path) and by the end of a shadow ray, consumed at ONE place, so that the ray set-up
    // code exists once and the traversal state is rewritten in one region of the loop only
    bool busy = false; int phase = 0; uint32_t path = 0; uint32_t flags = 0; int vis = 0; int start = 0;
    float edx = 0, edy = 0, edz = 0;
    LaneTrav L; L.cur = kDone; L.sp = 0; L.held = 0;
    for (;;) {
        if (next == end && !exhausted) {    // open the prefetched chunk and draw the one after it
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
        // ---- refill: lanes without a ray take the next paths of the chunk (kept wave-complete: no early exits above)
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
                if (ro.w > 0.0f) {  // pending NEE shadow ray [eps, tfar] (path_tracing.h:124-128)
                    const Rec4 rs = q.rs[path];
                    L.ray.dx = rs.x; L.ray.dy = rs.y; L.ray.dz = rs.z; L.ray.tfar = ro.w;
                    start = 1;
                } else if (!(flags & PF_NO_EXT)) start = 2;
                else { q.rh[path] = mk4(0.0f, 0.0f, 0.0f, u2f(0u)); busy = false; }
            }
            next += n_idle < left ? n_idle : left;
        }
        if (start != 0) {
            // shadow ray [eps, tfar]; extension ray [eps, inf), camera rays from 0 (camera.cpp:46, path_tracing.h:236)
            const bool ext = start == 2;
            if (ext) { L.ray.dx = edx; L.ray.dy = edy; L.ray.dz = edz; }
            trav_begin(L, (ext && (flags & 0xffffu) == 2u) ? 0.0f : sc.eps, ext ? INFINITY : L.ray.tfar);
            phase = ext ? 1 : 0; start = 0;
        }
        if (__ballot(busy) == 0ull) { if (exhausted && next == end) break; else continue; }
        if (STATS) { st_outer++; st_busy += __popcll(__ballot(busy)); }
        // ---- while-while traversal: descend inner nodes until every busy lane sits on a leaf (or is done) ...
        // (lanes that reach a leaf wait here; once only a few lanes are still descending, everybody moves on to the
        // leaf phase and the stragglers resume in the next round)
        for (;;) {
#if LJ_EXT_HOLD && LJ_EXT_POOL
            if (busy && L.cur < 0 && L.held == 0 && (LJ_EXT_HOLD_SHADOW || phase != 0)) trav_hold<RESIDENT>(tv, L);
#endif
            const bool descending = busy && L.cur >= 0 && L.cur != kDone;
            const unsigned long long dm = __ballot(descending);
            if (dm == 0ull) break;
            // only hand over to the leaf phase if some lane actually has a leaf to test (otherwise no progress is made)
            if ((uint32_t)__popcll(dm) < min_descending && __ballot(busy && (L.cur < 0 || L.held != 0)) != 0ull) break;
            if (STATS) { st_nodes++; st_node_lanes += __popcll(dm); }
            if (descending) trav_node_step<RESIDENT>(tv, L);
        }
#if LJ_EXT_POOL
        // ... then the wave tests the primitives of all those leaves together, 64 (ray, primitive) pairs at a time
        {
            uint32_t n_pairs;
            const uint32_t rounds = trav_leaf_pool<RESIDENT, SPHERES>(tv, lp, L, busy && (L.cur < 0 || L.held != 0), phase == 0, n_pairs);
            if (STATS) { st_leaf += rounds; st_leaf_lanes += n_pairs; }
        }
#else
        if (STATS) {
            const bool at_leaf = busy && L.cur < 0;
            int cnt = at_leaf ? ((~L.cur) & 7) + 1 : 0, mx = cnt, sum = cnt;
            for (int o = 32; o > 0; o >>= 1) { mx = max(mx, __shfl_xor(mx, o, 64)); sum += __shfl_xor(sum, o, 64); }
            st_leaf += mx; st_leaf_lanes += sum;
        }
        // ... then all of them test their leaf together
        if (busy && L.cur < 0) trav_leaf_step<RESIDENT, SPHERES>(tv, L, phase == 0);
#endif
        // ---- ray finished?
        if (STATS) st_rays += __popcll(__ballot(busy && L.cur == kDone));
        if (busy && L.cur == kDone) {
            if (phase == 0) {
                vis = (L.best.gprim < 0) ? HIT_VIS_BIT : 0;
                if (!(flags & PF_NO_EXT)) { start = 2; L.cur = 0; }   // (cur leaves kDone here: this block must not run twice)
                else { q.rh[path] = mk4(0.0f, 0.0f, 0.0f, u2f((uint32_t)vis)); busy = false; }
            } else {
                const uint32_t code = (uint32_t)vis | (uint32_t)(L.best.gprim + 1);
                const bool hit = L.best.gprim >= 0;
                trav_finish(L);
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

// ---------------------------------------------------------------- shade + compaction
struct ShadeSortBuf { uint32_t *perm; uint8_t *keys; uint32_t octant_bin; };   // (perm == nullptr: chunks are compacted in place)
// Where a chunk's survivors go inside the output range [out, out + total): in thread order, or — `octant_bin` — grouped by the direction
// octant of their new extension ray (a counting sort over eight keys by wave ballots), so that the 64 rays a wave of the extend kernel picks
// up start out in the same octant.  Returns the survivor's offset and the chunk's total (identical in every thread).  Call from all threads.
__device__ __forceinline__ uint32_t survivor_offset(bool alive, const PathState &ps, bool octant_bin, uint32_t (*s_cnt)[kBlock / 64], uint32_t &total) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (!octant_bin) {
        const unsigned long long mask = __ballot(alive);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (lane == 0u) s_cnt[0][wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t before = 0; total = 0;
        for (uint32_t w = 0; w < kBlock / 64; w++) { const uint32_t n = s_cnt[0][w]; before += (w < wave) ? n : 0u; total += n; }
        return before + rank;   // (the caller alternates between two count tables, so the next chunk's counts do not race these reads)
    }
    const uint32_t oct = (ps.dir.x < 0.0f ? 1u : 0u) | (ps.dir.y < 0.0f ? 2u : 0u) | (ps.dir.z < 0.0f ? 4u : 0u);
    uint32_t rank = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8u; k++) {
        const unsigned long long m = __ballot(alive && oct == k);
        if (alive && oct == k) rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane == 0u) s_cnt[k][wave] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    uint32_t before = 0; total = 0;
    for (uint32_t k = 0; k < 8u; k++) for (uint32_t w = 0; w < kBlock / 64; w++) {
        const uint32_t n = s_cnt[k][w];
        before += (k < oct || (k == oct && w < wave)) ? n : 0u; total += n;
    }
    return before + rank;
}

// What a path will do in this shade step, as far as its queue records tell — the key its chunk is sorted by before shading (below):
// 0 nothing (no extension ray was traced: only the pending next-event estimate is added), 1 its ray left the scene (environment map),
// 2 it only accounts for the hit (Russian roulette ended it), 3 + k full shading on Material alternative k (material.h:102-110).
constexpr int kShadeKeys = 12;
template <class Ft>
__device__ __forceinline__ int shade_key(const DScene &sc, const DQueue &q, uint32_t slot) {
    const uint32_t flags = f2u(q.rd[slot].w);
    if (flags & PF_NO_EXT) return 0;
    const int gprim = (int)(f2u(q.rh[slot].w) & 0x3fffffffu) - 1;
    if (gprim < 0) return 1;
    if (flags & PF_DYING) return 2;
    return 3 + sc.materials[sc.prims[gprim].material_id].kind;
}
// Scenes with several Material alternatives or an environment map: the lanes of a wave would each take another branch of shade_path
// (disney_bsdf.xml: 42 % of lanes active per instruction).  Such a feature set sorts every 256-path chunk by shade_key first.
template <class Ft> struct ShadeSorted { static constexpr bool value = Ft::envmap || (Ft::kinds & (Ft::kinds - 1u)) != 0u; };

// Shade the `count` live paths at the front of one segment chunk by chunk; survivors are compacted to the front, in
// order (stable).  Returns the number of survivors (identical in every thread).  s_wcnt: 2 x (kBlock / 64) words of LDS.
// Sorted feature sets: within a chunk, thread t takes the t-th path in (key, slot) order — a counting sort by wave ballots, one LDS
// table of counts and one of slots — so that a wave's lanes run the same branch of shade_path; the survivors are then compacted in that
// order.  A path's value does not depend on where it sits in the queue (its radiance goes to sample_rgb[sample]), so images are unchanged.
template <class Ft>
__device__ __forceinline__ uint32_t shade_compact_segment(const DScene &sc, const DPass &pass, const DQueue &q, uint32_t base, uint32_t count, ShadeCounters &cnt, uint32_t (*s_oct)[kBlock / 64], bool octant_bin = false) {
    constexpr bool SORT = ShadeSorted<Ft>::value;
    __shared__ uint16_t s_perm[SORT ? kBlock : 1];
    __shared__ uint16_t s_kcnt[SORT ? kShadeKeys : 1][kBlock / 64];
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t out = 0;  // survivors written so far (identical in every thread)
    for (uint32_t c0 = 0, it = 0; c0 < count; c0 += kBlock, it++) {
        uint32_t j = c0 + threadIdx.x;
        if (SORT) {
            const int key = j < count ? shade_key<Ft>(sc, q, base + j) : kShadeKeys;   // (slots beyond the live front sort last)
            uint32_t rank_in_key = 0;
#pragma unroll
            for (int k = 0; k < kShadeKeys; k++) {
                if (k >= 3 && !Ft::kind(k - 3)) continue;
                if (k == 1 && !Ft::envmap) { /* misses still exist without an environment map: they end the path */ }
                const unsigned long long m = __ballot(key == k);
                if (key == k) rank_in_key = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if ((threadIdx.x & 63) == 0) s_kcnt[k][wave] = (uint16_t)__popcll(m);
            }
            __syncthreads();
            if (key < kShadeKeys) {
                uint32_t pos = rank_in_key;
#pragma unroll
                for (int k = 0; k < kShadeKeys; k++) {
                    if (k >= 3 && !Ft::kind(k - 3)) continue;
                    for (uint32_t w = 0; w < kBlock / 64; w++) { const uint32_t n = s_kcnt[k][w]; pos += (k < key || (k == key && w < wave)) ? n : 0u; }
                }
                s_perm[pos] = (uint16_t)threadIdx.x;
            }
            __syncthreads();
            const uint32_t live = count - c0 < kBlock ? count - c0 : kBlock;
            j = threadIdx.x < live ? c0 + s_perm[threadIdx.x] : count;
        }
        bool alive = false;
        PathState ps;
        if (j < count) {
            q_load_for_shade(q, base + j, ps);
            alive = shade_path<Ft>(sc, pass, ps, cnt);
            if (!alive) {
                float *o = pass.sample_rgb + 3ull * ps.sample;
                o[0] = ps.rad.x; o[1] = ps.rad.y; o[2] = ps.rad.z;
                cnt.done++;
            }
        }
        // every record of this chunk has been read (and consumed) once the threads meet in survivor_offset, so writing into
        // [out, out + survivors) — which never reaches past the end of this chunk — cannot overtake a read
        uint32_t total;
        const uint32_t off = survivor_offset(alive, ps, octant_bin, s_oct + (it & 1u) * 8u, total);
        if (alive) q_store(q, base + out + off, ps);
        out += total;
    }
    return out;
}

// The same with the WHOLE live front of the segment sorted by shade_key, not each 256-path chunk of it (k_shade, scenes with several
// Material alternatives or an environment map).  Why: a chunk costs what its slowest wave costs — with the chunk's paths sorted that
// is still the wave that got the heaviest class (disney_bsdf.xml: the Disney BSDF on 15 % of the paths), while the other three wait
// at the chunk's barrier; sorted over the whole segment, most chunks hold ONE class and cost what that class costs.  Three passes over
// the front: (1) keys (one byte per path, kept in `sb.keys`) and their histogram, (2) a stable scatter of the slot numbers into
// `sb.perm` (key-major), (3) shading in that order — reading the queue records of slot perm[t] from `q` and writing the survivors,
// compacted, to `qo`: a SECOND set of queue records, because a path read from anywhere in the segment may not be overwritten by an
// earlier survivor.  The extend launch that follows works on `qo`; the two sets swap roles every step.
template <class Ft>
__device__ __forceinline__ uint32_t shade_sorted_segment(const DScene &sc, const DPass &pass, const DQueue &q, const DQueue &qo, const ShadeSortBuf &sb, uint32_t base, uint32_t count,
                                                         ShadeCounters &cnt, uint32_t (*s_oct)[kBlock / 64]) {
    __shared__ uint32_t s_start[kShadeKeys];
    __shared__ uint16_t s_kc[kShadeKeys][kBlock / 64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (threadIdx.x < kShadeKeys) s_start[threadIdx.x] = 0u;
    __syncthreads();
    // ---- (1) keys and histogram
    for (uint32_t c0 = 0; c0 < count; c0 += kBlock) {
        const uint32_t j = c0 + threadIdx.x;
        const int key = j < count ? shade_key<Ft>(sc, q, base + j) : kShadeKeys;
        if (j < count) sb.keys[base + j] = (uint8_t)key;
#pragma unroll
        for (int k = 0; k < kShadeKeys; k++) {
            if (k >= 3 && !Ft::kind(k - 3)) continue;
            const unsigned long long m = __ballot(key == k);
            if (lane == 0u && m != 0ull) atomicAdd(&s_start[k], (uint32_t)__popcll(m));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int k = 0; k < kShadeKeys; k++) { const uint32_t c = s_start[k]; s_start[k] = run; run += c; } }
    __syncthreads();
    // ---- (2) slot numbers in (key, slot) order
    for (uint32_t c0 = 0; c0 < count; c0 += kBlock) {
        const uint32_t j = c0 + threadIdx.x;
        const int key = j < count ? (int)sb.keys[base + j] : kShadeKeys;
        uint32_t rank_in_key = 0;
#pragma unroll
        for (int k = 0; k < kShadeKeys; k++) {
            if (k >= 3 && !Ft::kind(k - 3)) continue;
            const unsigned long long m = __ballot(key == k);
            if (key == k) rank_in_key = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (lane == 0u) s_kc[k][wave] = (uint16_t)__popcll(m);
        }
        __syncthreads();
        if (key < kShadeKeys) {
            uint32_t pos = s_start[key] + rank_in_key;
            for (uint32_t w = 0; w < wave; w++) pos += s_kc[key][w];
            sb.perm[base + pos] = j;
        }
        __syncthreads();
        if (threadIdx.x < kShadeKeys) {
            const int k = (int)threadIdx.x;
            if (k < 3 || Ft::kind(k - 3)) { uint32_t n = 0; for (uint32_t w = 0; w < kBlock / 64; w++) n += s_kc[k][w]; s_start[k] += n; }
        }
        __syncthreads();
    }
    // ---- (3) shade in that order; survivors compacted into the other record set
    uint32_t out = 0;
    for (uint32_t c0 = 0, it = 0; c0 < count; c0 += kBlock, it++) {
        const uint32_t t = c0 + threadIdx.x;
        bool alive = false;
        PathState ps;
        if (t < count) {
            const uint32_t j = sb.perm[base + t];
            q_load_for_shade(q, base + j, ps);
            alive = shade_path<Ft>(sc, pass, ps, cnt);
            if (!alive) {
                float *o = pass.sample_rgb + 3ull * ps.sample;
                o[0] = ps.rad.x; o[1] = ps.rad.y; o[2] = ps.rad.z;
                cnt.done++;
            }
        }
        uint32_t total;
        const uint32_t off = survivor_offset(alive, ps, sb.octant_bin != 0u, s_oct + (it & 1u) * 8u, total);
        if (alive) q_store(qo, base + out + off, ps);
        out += total;
    }
    return out;
}

// (the feature sets the shade kernel is compiled for — FeatLambert ... FeatAll — are listed in dshade.h)
#ifndef LJ_LAMBERT_OCC
#define LJ_LAMBERT_OCC 4
#endif
#ifndef LJ_SHADE_OCC
#define LJ_SHADE_OCC 3   // the feature sets beyond Lambert-only need 140 - 165 VGPRs: at 4 waves they spill (matpreview -3 % at 3)
#endif
#ifndef LJ_SHADE_OCC_LARGE
#define LJ_SHADE_OCC_LARGE 4   // ... but the two that run alone on the GPU beside a large tree's extend launches (one lane) gain from the fourth wave:
#endif                         // sponza 256 spp 200.0 -> 196.8 ms (shade alone -6.5 %), disney_bsdf 256 spp 91.6 -> 89.9 (tools/variant_ab.sh)
template <class Ft> struct ShadeOccupancy { static constexpr int waves = LJ_SHADE_OCC; };
template <> struct ShadeOccupancy<FeatLambert> { static constexpr int waves = LJ_LAMBERT_OCC; };
template <> struct ShadeOccupancy<FeatLambertTex> { static constexpr int waves = LJ_SHADE_OCC_LARGE; };
template <> struct ShadeOccupancy<FeatDisney> { static constexpr int waves = LJ_SHADE_OCC_LARGE; };

template <class Ft, int STAGE>
__global__ void __launch_bounds__(kBlock, ShadeOccupancy<Ft>::waves) k_shade(DScene sc, DPass pass, DQueue q, DQueue qo, ShadeSortBuf sb, DBlockState *blocks, uint32_t seg, ShadeStage stg, uint32_t *work, uint32_t *chunk_list, uint32_t parity, uint32_t extend_waves) {
    __shared__ uint32_t s_list_base;
    if (blockIdx.x == 0 && threadIdx.x == 0) work[0] = extend_waves;   // the extend launch that follows draws list entries beyond its own waves from it
    __shared__ uint32_t s_wcnt[16][kBlock / 64];   // two tables of (octant x wave) survivor counts, used alternately
    __shared__ unsigned long long s_cnt[5];
    stage_shade_tables<STAGE>(sc, stg, 0u);
    DBlockState &bs = blocks[blockIdx.x];
    const uint32_t count = bs.count, next_sample = bs.next_sample, end_sample = bs.end_sample;
    if (threadIdx.x < 5) s_cnt[threadIdx.x] = 0ull;
    __syncthreads();
    ShadeCounters cnt; cnt.bounces = cnt.closest = cnt.shadow = cnt.done = 0;
    const uint32_t base = blockIdx.x * seg;
    // ---- shade the live front of the segment chunk by chunk; survivors are compacted to the front, in order
    // (qo: the record set the survivors go to — `q` itself unless the segment is sorted as a whole, shade_sorted_segment)
    uint32_t out;
    if (ShadeSorted<Ft>::value && sb.perm != nullptr) out = shade_sorted_segment<Ft>(sc, pass, q, qo, sb, base, count, cnt, s_wcnt);
    else out = shade_compact_segment<Ft>(sc, pass, q, base, count, cnt, s_wcnt, sb.octant_bin != 0u);
    // ---- refill the rest of the segment with the workgroup's next camera samples (path_tracing.h:10-14)
    const uint32_t left = end_sample - next_sample, room = seg - out;
    const uint32_t n_new = left < room ? left : room;
    for (uint32_t g = threadIdx.x; g < n_new; g += kBlock) {
        PathState ps;
        generate_path(sc, pass, next_sample + g, ps);
        q_store(qo, base + out + g, ps);
    }
    const uint32_t b = wave_sum(cnt.bounces), cl = wave_sum(cnt.closest), sh = wave_sum(cnt.shadow), dn = wave_sum(cnt.done);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_cnt[0], (unsigned long long)b); atomicAdd(&s_cnt[1], (unsigned long long)cl); atomicAdd(&s_cnt[2], (unsigned long long)sh);
        atomicAdd(&s_cnt[3], (unsigned long long)dn);
    }
    __syncthreads();
    // list this segment's live chunks for the extend launch (one atomic per workgroup)
    const uint32_t live_chunks = (out + n_new + kChunk - 1) / kChunk;
    if (threadIdx.x == 0) s_list_base = live_chunks ? atomicAdd(&work[1 + parity], live_chunks) : 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < live_chunks; i += kBlock) chunk_list[s_list_base + i] = blockIdx.x * (seg / kChunk) + i;
    if (threadIdx.x == 0) {
        bs.next_sample = next_sample + n_new;
        bs.count = out + n_new;
        bs.bounce_iterations += s_cnt[0]; bs.rays_closest += s_cnt[1]; bs.rays_shadow += s_cnt[2]; bs.samples_done += s_cnt[3]; bs.path_steps += count;
    }
}

// ---------------------------------------------------------------- the tail of a render, fused
// When no camera sample is left to start and only the long-lived paths remain, a launch pair per bounce is mostly launch
// overhead (~10 us each, tens of bounces).  A workgroup's segment is independent of every other one, so the rest of the
// render runs inside ONE launch: each workgroup alternates "trace my live paths" and "shade + compact my segment" until
// its segment is empty.  Per-sample values are the same as with separate launches, bit for bit.
template <class Ft>
__global__ void __launch_bounds__(kBlock, 2) k_tail(DScene sc, DPass pass, DQueue q, DBlockState *blocks, uint32_t seg, ShadeStage stg, uint32_t shade_lds_at,
                                                    int stack, int lds_nodes, int lds_prims, int *spill) {
    __shared__ uint32_t s_wcnt[16][kBlock / 64];   // two tables of (octant x wave) survivor counts, used alternately
    const TreeView tv = stage_tree(sc, stack, lds_nodes, lds_prims, spill, gridDim.x * kBlock, blockIdx.x * kBlock + threadIdx.x);
    DScene ssc = sc;                       // the shading view of the scene: tables in LDS
    stage_shade_tables<-1>(ssc, stg, shade_lds_at);
    __syncthreads();
    DBlockState &bs = blocks[blockIdx.x];
    uint32_t count = bs.count;
    const uint32_t base = blockIdx.x * seg;
    ShadeCounters cnt; cnt.bounces = cnt.closest = cnt.shadow = cnt.done = 0;
    unsigned long long path_steps = 0;
    for (int guard = 0; guard < (1 << 16) && count > 0; guard++) {
        // ---- extend: one lane per path, shadow ray (any hit) then extension ray (closest hit)
        for (uint32_t i = threadIdx.x; i < count; i += kBlock) {
            const uint32_t path = base + i;
            const Rec4 ro = q.ro[path], rd = q.rd[path];
            const uint32_t flags = f2u(rd.w);
            LaneTrav L;
            L.ray.ox = ro.x; L.ray.oy = ro.y; L.ray.oz = ro.z;
            uint32_t vis = 0;
            if (ro.w > 0.0f) {
                const Rec4 rs = q.rs[path];
                L.ray.dx = rs.x; L.ray.dy = rs.y; L.ray.dz = rs.z;
                trav_begin(L, sc.eps, ro.w);
                while (L.cur != kDone) {
                    while (L.cur >= 0 && L.cur != kDone) trav_node_step<false>(tv, L);
                    if (L.cur < 0) trav_leaf_step<false, true>(tv, L, true);
                }
                vis = (L.best.gprim < 0) ? (uint32_t)HIT_VIS_BIT : 0u;
            }
            if (!(flags & PF_NO_EXT)) {
                L.ray.dx = rd.x; L.ray.dy = rd.y; L.ray.dz = rd.z;
                trav_begin(L, ((flags & 0xffffu) == 2u) ? 0.0f : sc.eps, INFINITY);
                while (L.cur != kDone) {
                    while (L.cur >= 0 && L.cur != kDone) trav_node_step<false>(tv, L);
                    if (L.cur < 0) trav_leaf_step<false, true>(tv, L, false);
                }
                trav_finish(L);
                const bool hit = L.best.gprim >= 0;
                q.rh[path] = mk4(hit ? L.best.t : 0.0f, L.best.u, L.best.v, u2f(vis | (uint32_t)(L.best.gprim + 1)));
            } else q.rh[path] = mk4(0.0f, 0.0f, 0.0f, u2f(vis));
        }
        __syncthreads();   // (workgroup-scope release/acquire of the records just written)
        // ---- shade + compact
        path_steps += count;
        count = shade_compact_segment<Ft>(ssc, pass, q, base, count, cnt, s_wcnt);
        __syncthreads();
    }
    __shared__ unsigned long long s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0ull;
    __syncthreads();
    const uint32_t b = wave_sum(cnt.bounces), cl = wave_sum(cnt.closest), sh = wave_sum(cnt.shadow), dn = wave_sum(cnt.done);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_cnt[0], (unsigned long long)b); atomicAdd(&s_cnt[1], (unsigned long long)cl); atomicAdd(&s_cnt[2], (unsigned long long)sh);
        atomicAdd(&s_cnt[3], (unsigned long long)dn);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bs.count = count;
        bs.bounce_iterations += s_cnt[0]; bs.rays_closest += s_cnt[1]; bs.rays_shadow += s_cnt[2]; bs.samples_done += s_cnt[3]; bs.path_steps += path_steps;
    }
}

// ---------------------------------------------------------------- resolve: radiance / spp, deterministic
// One wave per pixel; lanes stride over the pixel's samples, then a fixed xor-butterfly combines the 64 partials.
__global__ void __launch_bounds__(kBlock) k_resolve(DPass pass, uint32_t n_pixels, float *rgb) {
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= n_pixels) return;
    const float *src = pass.sample_rgb + 3ull * (uint64_t)wave * pass.spp;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t s = lane; s < pass.spp; s += 64) { r += src[3 * s]; g += src[3 * s + 1]; b += src[3 * s + 2]; }
    for (int o = 32; o > 0; o >>= 1) { r += __shfl_xor(r, o, 64); g += __shfl_xor(g, o, 64); b += __shfl_xor(b, o, 64); }
    if (lane == 0) {
        const uint32_t pixel = pass.pixel_list[wave];
        const float inv = div_ieee(1.0f, (float)pass.spp);
        rgb[3ull * pixel] = r * inv; rgb[3ull * pixel + 1] = g * inv; rgb[3ull * pixel + 2] = b * inv;
    }
}

// ---------------------------------------------------------------- batched ray queries for the parity tests

// Closest hit (or any hit) of every lane's ray, the whole wave together, with k_extend's two phases — inner nodes lane by lane while enough
// lanes descend, leaves pooled.  L: set up by trav_begin (lanes without a ray: L.cur = kDone).  Ends with trav_finish.
__device__ __forceinline__ void trace_wave(const TreeView &tv, const LeafPool &lp, LaneTrav &L, const bool any_hit) {
#if LJ_EXT_POOL
    for (;;) {
        for (;;) {
#if LJ_EXT_HOLD && LJ_EXT_POOL
            if (L.cur < 0 && L.held == 0) trav_hold<false>(tv, L);
#endif
            const bool descending = L.cur >= 0 && L.cur != kDone;
            if (__ballot(descending) == 0ull) break;
            if (descending) trav_node_step<false>(tv, L);
        }
        const bool at_leaf = L.cur < 0 || L.held != 0;
        if (__ballot(at_leaf) == 0ull) break;
        uint32_t n_pairs;
        (void)trav_leaf_pool<false, true>(tv, lp, L, at_leaf, any_hit, n_pairs);
    }
#else
    while (L.cur != kDone) {
        while (L.cur >= 0 && L.cur != kDone) trav_node_step<false>(tv, L);
        if (L.cur < 0) trav_leaf_step<false, true>(tv, L, any_hit);
    }
#endif
    trav_finish(L);
}

// (wave-complete iterations over the rays, traced by trace_wave: the parity tests of intersect() / occluded() hold the pooled leaf phase
// to the oracle, bit for bit)
__global__ void __launch_bounds__(kBlock) k_trace_rays(DScene sc, const RayIO *rays, long long n, HitIO *hits, unsigned char *occ, int stack, int lds_nodes, int lds_prims, int *spill, uint32_t pool_at) {
    const TreeView tv = stage_tree(sc, stack, lds_nodes, lds_prims, spill, gridDim.x * kBlock, blockIdx.x * kBlock + threadIdx.x);
    LeafPool lp{};
#if LJ_EXT_POOL
    lp = leaf_pool_at(pool_at);
#endif
    for (long long i0 = (long long)blockIdx.x * kBlock; i0 < n; i0 += (long long)gridDim.x * kBlock) {
        const long long i = i0 + threadIdx.x;
        const bool act = i < n;
        LaneTrav L;
        L.ray.ox = 0.0f; L.ray.oy = 0.0f; L.ray.oz = 0.0f; L.ray.dx = 0.0f; L.ray.dy = 0.0f; L.ray.dz = 1.0f;
        if (act) {
            L.ray.ox = rays[i].org[0]; L.ray.oy = rays[i].org[1]; L.ray.oz = rays[i].org[2];
            L.ray.dx = rays[i].dir[0]; L.ray.dy = rays[i].dir[1]; L.ray.dz = rays[i].dir[2];
        }
        trav_begin(L, act ? rays[i].tnear : 0.0f, act ? rays[i].tfar : 0.0f);
        if (!act) L.cur = kDone;
        trace_wave(tv, lp, L, occ != nullptr);
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

// ---------------------------------------------------------------- auxiliary buffers (render.cpp:12-69)
// One thread per listed pixel: primary ray through the pixel centre, closest hit, aux_value().
__global__ void __launch_bounds__(kBlock) k_aux(DScene sc, const uint32_t *pixel_list, uint32_t n_pixels, int integrator, float *rgb, int stack, int lds_nodes, int lds_prims, int *spill) {
    const TreeView tv = stage_tree(sc, stack, lds_nodes, lds_prims, spill, gridDim.x * kBlock, blockIdx.x * kBlock + threadIdx.x);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_pixels; i += gridDim.x * kBlock) {
        const uint32_t pixel = pixel_list[i];
        const int x = (int)(pixel % (uint32_t)sc.cam.width), y = (int)(pixel / (uint32_t)sc.cam.width);
        const f3 org = ld3(sc.cam.org), dir = camera_primary_dir(sc.cam, x, y, 0.5f, 0.5f);
        LaneTrav L;
        L.ray.ox = org.x; L.ray.oy = org.y; L.ray.oz = org.z; L.ray.dx = dir.x; L.ray.dy = dir.y; L.ray.dz = dir.z;
        trav_begin(L, 0.0f, INFINITY);   // camera.cpp:46: tnear 0
        while (L.cur != kDone) {
            while (L.cur >= 0 && L.cur != kDone) trav_node_step<false>(tv, L);
            if (L.cur < 0) trav_leaf_step<false, true>(tv, L, false);
        }
        trav_finish(L);
        const f3 c = aux_value(sc, integrator, org, dir, L.best.t, L.best.u, L.best.v, L.best.gprim);
        rgb[3ull * pixel] = c.x; rgb[3ull * pixel + 1] = c.y; rgb[3ull * pixel + 2] = c.z;
    }
}

// ---------------------------------------------------------------- launchers (called from api_device.hip)
// A traversal of a BVH4 with `depth` inner levels holds at most 3 * depth entries.  The first `stack` levels of every
// lane's stack live in LDS (1 KiB per level and workgroup); deeper levels — rare — go to a global overflow buffer.
// LDS per 256-thread workgroup = stack * 1 KiB + staged nodes * 112 B + staged prims * 48 B; four workgroups share a
// CU's 160 KiB, so the budget per workgroup is 40 KiB.

ExtendConfig extend_config(int n_nodes, int n_prims, int bvh_depth, int n_spheres, int n_nodes8, int bvh8_depth, bool open_scene) {
    ExtendConfig c{};
    c.spheres = n_spheres > 0 ? 1 : 0;
    const int need = 3 * (bvh_depth < 1 ? 1 : bvh_depth);
    // LDS image: small scenes (which may become fully resident) get 16 stack levels and up to 40 KiB; for the others 12
    // levels + 20 KiB (the top ~70 nodes): with the 10 KiB of leaf pools a workgroup then takes 30 KiB and five fit a CU (the kernel
    // needs 92 VGPRs: five waves per SIMD).  tools/lds_sweep.sh, 64 spp: disney_bsdf 28.5 ms at 20-22 KiB against 29.8 at 24 (four
    // workgroups), sponza flat (63.5 / 63.8); 14 KiB: +2 %; 64 KiB: +30 % — it crowds out the other workgroups of the CU
    int cap = n_prims <= 256 ? 16 : 12, kib = n_prims <= 256 ? (LJ_EXT_POOL ? 32 : 40) : 20;
    if (const char *e = getenv("LJ_TUNE_EXT_STACK")) cap = atoi(e);
    if (const char *e = getenv("LJ_TUNE_EXT_LDS_KB")) kib = atoi(e);
    c.stack = need < cap ? need : cap;
    c.spill_levels = need - c.stack;
    const int budget = kib * 1024 - 256 - c.stack * kBlock * 4;
    const int small = n_prims <= 256;                                     // primitives only when the whole scene fits (<= 12 KiB)
    c.lds_prims = small ? n_prims : 0;
    int max_nodes = (budget - c.lds_prims * 48) / 112;
    if (max_nodes < 0) max_nodes = 0;
    c.lds_nodes = n_nodes < max_nodes ? n_nodes : max_nodes;
    if (!LJ_EXT_LDS_NODES && c.lds_nodes < n_nodes) c.lds_nodes = 0;
    // whole tree, all primitives and every stack level (+1: the branch-free pushes store one slot ahead) in LDS
    c.resident = (c.lds_nodes == n_nodes && c.lds_prims == n_prims && c.spill_levels == 0 && c.stack + 1 <= 16) ? 1 : 0;
    if (c.resident) c.stack += 1;
    c.smem = (size_t)c.stack * kBlock * 4 + (size_t)c.lds_nodes * 112 + (size_t)c.lds_prims * 48;
    // tuned on MI355X (tools/pool_probe.sh): when the tree is LDS-resident a node step is cheap and waiting for the last
    // descending lane costs little; with nodes in L2 the (pooled) leaf phase starts once fewer than 32 lanes still descend
    // (sponza / disney_bsdf at 64 spp: 16: 65.2 / 30.6 ms, 24: 64.2 / 30.0, 32: 63.2 / 29.7, 40: 63.3 / 30.3, 48: 67.8 / 31.9; with held
    // leaves — a lane that reaches a leaf keeps descending until its second one — 24: 62.6 / 27.8, 32: 61.7 / 27.5, 40: 60.9 / 27.4)
    c.refill_min = 8; c.min_descending = (n_nodes <= c.lds_nodes) ? 1 : (LJ_EXT_POOL ? (LJ_EXT_HOLD ? 40 : 32) : 24);
    // A tree beyond the LDS image can be traversed as a BVH8 instead (k_extend8, extend8.hip): 8-byte group entries, at most one push per
    // step, 80-byte nodes.  Measured on MI355X (tools/bvh8_ab.sh, tools/ab1024.sh, profiles/r03_bvh8_ab.txt): 26 % fewer node steps per ray
    // and a third fewer vector-memory instructions, but 26 % more VALU instructions (eight quantised children cost ~200 per step), and with
    // five waves per SIMD the extend kernel is bound by instruction issue, not by the gather path.  Which tree wins depends on what the rays
    // do.  In a closed scene every ray ends on a surface and the BVH4's distance-sorted descent culls most of what lies behind it: sponza
    // 1024 spp 784 ms (BVH4) against 815.  In an open scene under an environment map most rays leave the scene — a ray that hits nothing
    // visits everything along its way whatever the order, so the BVH8's fewer steps are all gain: disney_bsdf 256 spp 82.5 ms against 85.3,
    // matpreview 64 spp 61.0 against 65.3, disney_metal 25.3 against 26.4.  So: BVH8 for trees beyond the LDS image of scenes lit by an
    // environment map, BVH4 otherwise; LJ_TUNE_BVH8=0 / 1 overrides.
    // A ray's stack is rarely more than four groups deep (sponza: 1 push in 600 lands on level 4, 1 in 10^5 on level 6), so six levels
    // live in LDS and the rest of the tree's depth goes to the global overflow buffer; with the first 96 nodes (three full levels and part
    // of the fourth) and the pools a workgroup takes 29.5 KiB: five per CU.  `spill_levels` counts 4-byte units per lane (ensure_spill):
    // two per group level.
    const bool can_wide = !c.resident && n_nodes > c.lds_nodes && n_nodes8 > 0;
    c.wide = (can_wide && open_scene) ? 1 : 0;
    if (const char *e = getenv("LJ_TUNE_BVH8")) c.wide = (atoi(e) != 0 && can_wide) ? 1 : 0;
    if (c.wide) {
        int cap8 = 6, nodes8 = 96;
        if (const char *e = getenv("LJ_TUNE_EXT8_STACK")) cap8 = atoi(e);
        if (const char *e = getenv("LJ_TUNE_EXT8_NODES")) nodes8 = atoi(e);
        const int need8 = bvh8_depth < 1 ? 1 : bvh8_depth;
        c.stack8 = need8 < cap8 ? need8 : cap8;
        if (c.stack8 < 1) c.stack8 = 1;
        c.lds_nodes8 = n_nodes8 < nodes8 ? n_nodes8 : nodes8;
        c.smem8 = (size_t)c.stack8 * kBlock * 8 + (size_t)c.lds_nodes8 * 80;
        const int spill8 = 2 * (need8 - c.stack8 > 0 ? need8 - c.stack8 : 0);
        if (spill8 > c.spill_levels) c.spill_levels = spill8;   // (k_aux / k_volpath still walk the BVH4 with the 4-byte levels)
    }
    return c;
}
int max_stack_depth() { return 40; }  // inner levels; the builder's own cap is 38

// LDS staging plan of the shade kernel; sizes are rounded up to 16 bytes (the device buffers are padded accordingly).
ShadeConfig shade_config(size_t n_prims, size_t n_materials, size_t n_lights, size_t n_light_tris, size_t n_light_tri_cdf, size_t n_images3, size_t n_images1, size_t n_env_marg) {
    auto r16 = [](size_t b) { return (uint32_t)((b + 15) & ~(size_t)15); };
    ShadeConfig c{};
    c.variant = kShadeVariantAll;
    c.materials_bytes = r16(n_materials * sizeof(DMaterial)); c.lights_bytes = r16(n_lights * sizeof(DLight));
    c.light_cdf_bytes = r16((n_lights + 1) * 4); c.light_tris_bytes = r16(n_light_tris * sizeof(DLightTri)); c.light_tri_cdf_bytes = r16(n_light_tri_cdf * 4);
    size_t small = (size_t)c.materials_bytes + c.lights_bytes + c.light_cdf_bytes + c.light_tris_bytes + c.light_tri_cdf_bytes;
    if (small > 24 * 1024) {  // too many materials / emissive triangles: leave everything in global memory
        c = ShadeConfig{}; c.variant = kShadeVariantAll; c.smem = 0; return c;
    }
    c.prims_bytes = r16(n_prims * sizeof(DPrimShade));
    c.stage_prims = (small + c.prims_bytes <= 32 * 1024) ? 1u : 0u;
    if (!c.stage_prims) c.prims_bytes = 0;
    c.smem = small + c.prims_bytes;
    // beside those, while they stay small: the image descriptors (the first, dependent, load of every texture lookup) and the
    // environment map's marginal cdf / pdf / guide (its first search then never leaves the CU)
    const uint32_t img3 = r16(n_images3 * sizeof(DImage)), img1 = r16(n_images1 * sizeof(DImage)), marg = r16(n_env_marg * 4);
    if (!(getenv("LJ_TUNE_STAGE_IMAGES") && atoi(getenv("LJ_TUNE_STAGE_IMAGES")) == 0)) {
        if (img3 + img1 <= 8 * 1024) { c.images3_bytes = img3; c.images1_bytes = img1; c.smem += img3 + img1; }
        if (n_env_marg > 0 && marg <= 8 * 1024) { c.env_marg_bytes = marg; c.smem += marg; }
    }
    return c;
}

// LDS of an extend launch: the traversal image, then the leaf pools of its four waves
size_t extend_smem(const ExtendConfig &cfg) { return ((cfg.smem + 15) & ~(size_t)15) + (LJ_EXT_POOL ? (kBlock / 64) * kWavePoolBytes : 0); }
void launch_extend(const DScene &sc, const DQueue &q, const DBlockState *blocks, uint32_t grid, uint32_t seg, uint32_t *work, const uint32_t *chunk_list, uint32_t parity, const ExtendConfig &cfg, int *spill, unsigned long long *stats, hipStream_t s) {
    if (cfg.wide) { launch_extend8(sc, q, blocks, grid, seg, work, chunk_list, parity, cfg, spill, stats, s); return; }
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), extend_smem(cfg), s, sc, q, blocks, seg, work, chunk_list, parity, cfg.stack, cfg.lds_nodes, cfg.lds_prims, spill, cfg.refill_min, cfg.min_descending, stats,
                           (uint32_t)((cfg.smem + 15) & ~(size_t)15));
    };
    const int variant = (stats ? 4 : 0) | (cfg.resident ? 2 : 0) | (cfg.spheres ? 1 : 0);
    switch (variant) {
        case 0: launch(k_extend<false, false, false>); break;
        case 1: launch(k_extend<false, false, true>); break;
        case 2: launch(k_extend<false, true, false>); break;
        case 3: launch(k_extend<false, true, true>); break;
        case 4: launch(k_extend<true, false, false>); break;
        case 5: launch(k_extend<true, false, true>); break;
        case 6: launch(k_extend<true, true, false>); break;
        default: launch(k_extend<true, true, true>); break;
    }
}
// shade_variant() returns the first (smallest) feature set that covers a scene
int shade_variant(uint32_t kinds, bool textured, bool envmap, bool sphere_lights) {
    for (int v = 0; v < kNumShadeVariants; v++) if (variant_covers(v, kinds, textured, envmap, sphere_lights)) return v;
    return kNumShadeVariants - 1;
}

bool shade_sorts_segments(const ShadeConfig &cfg) {   // feature sets whose k_shade sorts a segment as a whole (they need the second record set + sort buffers)
    if (const char *e = getenv("LJ_TUNE_SHADE_SORT")) { if (atoi(e) < 2) return false; }
    bool sorted = false;
    if (cfg.smem == 0) return ShadeSorted<FeatAll>::value;
    with_shade_variant(cfg.variant, [&](auto ft) { sorted = ShadeSorted<decltype(ft)>::value; });
    return sorted;
}
void launch_shade(const DScene &sc, const DPass &pass, const DQueue &q, const DQueue &qo, uint32_t *sort_perm, uint8_t *sort_keys, DBlockState *blocks, uint32_t n_blocks, uint32_t seg, const ShadeConfig &cfg, uint32_t *work, uint32_t *chunk_list, uint32_t parity, uint32_t extend_waves, hipStream_t s) {
    ShadeSortBuf sb; sb.perm = sort_perm; sb.keys = sort_keys;
    sb.octant_bin = (getenv("LJ_TUNE_OCTANT_BIN") && atoi(getenv("LJ_TUNE_OCTANT_BIN")) != 0) ? 1u : 0u;
    const ShadeStage st = make_shade_stage(cfg);
    auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(n_blocks), dim3(kBlock), cfg.smem, s, sc, pass, q, qo, sb, blocks, seg, st, work, chunk_list, parity, extend_waves); };
    // (a scene whose tables are not staged at all runs the all-features instantiation: lj_scene_upload picks it)
    const int stage = cfg.smem == 0 ? 0 : (cfg.stage_prims ? 2 : 1);
    if (stage == 0) { launch(k_shade<FeatAll, 0>); return; }
    with_shade_variant(cfg.variant, [&](auto ft) {
        using Ft = decltype(ft);
        if (stage == 2) launch(k_shade<Ft, 2>); else launch(k_shade<Ft, 1>);
    });
}
// LDS the fused tail needs: the extend image followed by the shade tables; 0 when that does not fit one workgroup's share
size_t tail_smem(const ExtendConfig &ecfg, const ShadeConfig &scfg) {
    const size_t at = (ecfg.smem + 15) & ~(size_t)15, total = at + scfg.smem;
    return total <= 64 * 1024 ? total : 0;
}
void launch_tail(const DScene &sc, const DPass &pass, const DQueue &q, DBlockState *blocks, uint32_t n_blocks, uint32_t seg, const ExtendConfig &ecfg, const ShadeConfig &scfg, int *spill, hipStream_t s) {
    const ShadeStage st = make_shade_stage(scfg);
    const uint32_t at = (uint32_t)((ecfg.smem + 15) & ~(size_t)15);
    const size_t smem = tail_smem(ecfg, scfg);
    auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(n_blocks), dim3(kBlock), smem, s, sc, pass, q, blocks, seg, st, at, ecfg.stack, ecfg.lds_nodes, ecfg.lds_prims, spill); };
    with_shade_variant(scfg.variant, [&](auto ft) { launch(k_tail<decltype(ft)>); });
}
void launch_resolve(const DPass &pass, uint32_t n_pixels, float *rgb, hipStream_t s) {
    const uint32_t waves_per_block = kBlock / 64;
    const uint32_t grid = (n_pixels + waves_per_block - 1) / waves_per_block;
    if (grid) hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(kBlock), 0, s, pass, n_pixels, rgb);
}
void launch_aux(const DScene &sc, const uint32_t *pixel_list, uint32_t n_pixels, int integrator, float *rgb, const ExtendConfig &cfg, int *spill, int grid, hipStream_t s) {
    if (n_pixels) hipLaunchKernelGGL(k_aux, dim3(grid), dim3(kBlock), cfg.smem, s, sc, pixel_list, n_pixels, integrator, rgb, cfg.stack, cfg.lds_nodes, cfg.lds_prims, spill);
}
void launch_trace_rays(const DScene &sc, const void *rays, long long n, void *hits, unsigned char *occ, const ExtendConfig &cfg, int *spill, int grid, hipStream_t s) {
    if (cfg.wide) { launch_trace_rays8(sc, rays, n, hits, occ, cfg, spill, grid, s); return; }
    hipLaunchKernelGGL(k_trace_rays, dim3(grid), dim3(kBlock), extend_smem(cfg), s, sc, (const RayIO *)rays, n, (HitIO *)hits, occ, cfg.stack, cfg.lds_nodes, cfg.lds_prims, spill, (uint32_t)((cfg.smem + 15) & ~(size_t)15));
}

} // namespace ljd
