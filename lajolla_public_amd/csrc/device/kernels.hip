// gfx950 kernels of the wavefront integrator.  Written for wave64 / 256-thread workgroups / 160 KB LDS per CU.
//
//   k_prepare   1 thread     : per-step bookkeeping entirely on the device (no host round trip per bounce)
//   k_generate  grid-stride  : camera samples -> free slots of the current queue      (path_tracing.h:10-14)
//   k_extend    persistent   : closest hit of the extension ray + any hit of the pending shadow ray over the
//                              flattened BVH; top of the tree and the per-lane traversal stacks live in LDS
//                                                                                      (intersection.cpp:7-85)
//   k_shade     grid-stride  : hit accounting, NEE, BSDF sampling, Russian roulette; survivors are compacted into
//                              the other queue with one wave-ballot + one atomic per wave (path_tracing.h:58-322)
//   k_resolve   wave/pixel   : fixed-order sum of the per-sample radiance -> radiance / spp (render.cpp:94)
//   k_trace_rays             : batched intersect()/occluded() for the parity tests
#include <hip/hip_runtime.h>
#include "dshade.h"
#include "dtrace.h"

namespace ljd {

constexpr int kBlock = 256;

// ---------------------------------------------------------------- queue <-> registers
__device__ __forceinline__ void q_load_for_shade(const DQueue &q, uint32_t i, PathState &ps) {
    ps.org = mk3(q.ox[i], q.oy[i], q.oz[i]); ps.dir = mk3(q.dx[i], q.dy[i], q.dz[i]);
    ps.ht = q.ht[i]; ps.hu = q.hu[i]; ps.hv = q.hv[i]; ps.hcode = q.hprim[i];
    ps.W = mk3(q.wr[i], q.wg[i], q.wb[i]); ps.rr = q.rr[i]; ps.p2 = q.p2[i];
    ps.rad = mk3(q.lr[i], q.lg[i], q.lb[i]); ps.nee = mk3(q.nr[i], q.ng[i], q.nb[i]);
    ps.sample = q.sample[i]; ps.rng = q.rng[i];
    ps.eta_scale = q.eta_scale[i]; ps.spread = q.spread[i]; ps.flags = q.flags[i];
    ps.sdir = mk3(0, 0, 0); ps.stfar = 0.0f;
}
__device__ __forceinline__ void q_store(const DQueue &q, uint32_t i, const PathState &ps) {
    q.ox[i] = ps.org.x; q.oy[i] = ps.org.y; q.oz[i] = ps.org.z;
    q.dx[i] = ps.dir.x; q.dy[i] = ps.dir.y; q.dz[i] = ps.dir.z;
    q.sx[i] = ps.sdir.x; q.sy[i] = ps.sdir.y; q.sz[i] = ps.sdir.z; q.st[i] = ps.stfar;
    q.wr[i] = ps.W.x; q.wg[i] = ps.W.y; q.wb[i] = ps.W.z; q.rr[i] = ps.rr; q.p2[i] = ps.p2;
    q.lr[i] = ps.rad.x; q.lg[i] = ps.rad.y; q.lb[i] = ps.rad.z;
    q.nr[i] = ps.nee.x; q.ng[i] = ps.nee.y; q.nb[i] = ps.nee.z;
    q.sample[i] = ps.sample; q.rng[i] = ps.rng;
    q.eta_scale[i] = ps.eta_scale; q.spread[i] = ps.spread; q.flags[i] = ps.flags;
}

// ---------------------------------------------------------------- step bookkeeping
__global__ void k_prepare(DCtrl *c) {
    const uint32_t n_in = c->n_out;  // survivors written by the previous shade into what is now the current queue
    const uint64_t remaining = c->total_samples - c->next_sample;
    const uint32_t room = c->capacity - n_in;
    const uint32_t n_new = (uint32_t)(remaining < (uint64_t)room ? remaining : (uint64_t)room);
    c->gen_base = c->next_sample; c->gen_offset = n_in; c->n_new = n_new;
    c->next_sample += n_new;
    c->n_in = n_in + n_new; c->n_out = 0;
    c->steps += (n_in + n_new) ? 1u : 0u;
    c->path_steps += n_in + n_new;
}

__global__ void __launch_bounds__(kBlock) k_generate(DScene sc, DPass pass, DQueue q, const DCtrl *c) {
    const uint32_t n_new = c->n_new, off = c->gen_offset;
    const uint64_t base = c->gen_base;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_new; i += gridDim.x * kBlock) {
        PathState ps;
        generate_path(sc, pass, (uint32_t)(base + i), ps);
        q_store(q, off + i, ps);
    }
}

// ---------------------------------------------------------------- extend: BVH traversal with LDS-staged tree top
template <int LDS_NODES, int LDS_PRIMS, int STACK>
struct LdsMem {
    const DNode *gnodes; const DPrim *gprims; const DSphere *spheres;
    const DNode *lnodes; const DPrim *lprims;
    int n_lnodes, n_lprims;
    int *stack;  // this lane's column: stack[level * kBlock]
    __device__ __forceinline__ DNode node(int i) const { return i < n_lnodes ? lnodes[i] : gnodes[i]; }
    __device__ __forceinline__ DPrim prim(int i) const { return i < n_lprims ? lprims[i] : gprims[i]; }
    __device__ __forceinline__ const DSphere &sphere(int s) const { return spheres[s]; }
    __device__ __forceinline__ void push(int sp, int v) { stack[sp * kBlock] = v; }
    __device__ __forceinline__ int pop(int sp) const { return stack[sp * kBlock]; }
    __device__ __forceinline__ int max_stack() const { return STACK; }
};

template <int LDS_NODES, int LDS_PRIMS, int STACK>
__device__ __forceinline__ void stage_tree(const DScene &sc, DNode *lnodes, DPrim *lprims, int &n_lnodes, int &n_lprims) {
    // cooperative copy, 16 bytes per lane per step (nodes are stored breadth-first, so a prefix is the top of the tree)
    n_lnodes = sc.n_nodes < LDS_NODES ? sc.n_nodes : LDS_NODES;
    n_lprims = sc.n_prims <= LDS_PRIMS ? sc.n_prims : 0;  // primitives only when the whole scene fits
    const float4 *src = reinterpret_cast<const float4 *>(sc.nodes);
    float4 *dst = reinterpret_cast<float4 *>(lnodes);
    for (int i = threadIdx.x; i < n_lnodes * 4; i += kBlock) dst[i] = src[i];
    src = reinterpret_cast<const float4 *>(sc.leaf_prims); dst = reinterpret_cast<float4 *>(lprims);
    for (int i = threadIdx.x; i < n_lprims * 3; i += kBlock) dst[i] = src[i];
    __syncthreads();
}

template <int LDS_NODES, int LDS_PRIMS, int STACK>
__global__ void __launch_bounds__(kBlock) k_extend(DScene sc, DQueue q, DCtrl *c) {
    __shared__ __attribute__((aligned(16))) DNode lnodes[LDS_NODES];
    __shared__ __attribute__((aligned(16))) DPrim lprims[LDS_PRIMS];
    __shared__ int lstack[STACK * kBlock];
    LdsMem<LDS_NODES, LDS_PRIMS, STACK> mem;
    stage_tree<LDS_NODES, LDS_PRIMS, STACK>(sc, lnodes, lprims, mem.n_lnodes, mem.n_lprims);
    mem.gnodes = sc.nodes; mem.gprims = sc.leaf_prims; mem.spheres = sc.spheres;
    mem.lnodes = lnodes; mem.lprims = lprims; mem.stack = lstack + threadIdx.x;
    const uint32_t n = c->n_in;
    uint32_t n_closest = 0, n_shadow = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        RayF ray;
        ray.ox = q.ox[i]; ray.oy = q.oy[i]; ray.oz = q.oz[i];
        const float stfar = q.st[i];
        const uint32_t flags = q.flags[i];
        int code = 0;
        if (stfar > 0.0f) {  // pending NEE shadow ray [eps, tfar] (path_tracing.h:124-128)
            ray.dx = q.sx[i]; ray.dy = q.sy[i]; ray.dz = q.sz[i]; ray.tnear = sc.eps; ray.tfar = stfar;
            HitRec h;
            if (!traverse<true>(mem, ray, h)) code |= HIT_VIS_BIT;
            n_shadow++;
        }
        float t = 0.0f, u = 0.0f, v = 0.0f;
        if (!(flags & PF_NO_EXT)) {  // extension ray [eps, inf) — camera rays start at 0 (camera.cpp:46, path_tracing.h:236)
            ray.dx = q.dx[i]; ray.dy = q.dy[i]; ray.dz = q.dz[i];
            ray.tnear = ((flags & 0xffffu) == 2u) ? 0.0f : sc.eps; ray.tfar = INFINITY;
            HitRec h;
            if (traverse<false>(mem, ray, h)) { code |= (h.gprim + 1); t = h.t; u = h.u; v = h.v; }
            n_closest++;
        }
        q.ht[i] = t; q.hu[i] = u; q.hv[i] = v; q.hprim[i] = code;
    }
    (void)n_closest; (void)n_shadow;
}

// ---------------------------------------------------------------- shade + compaction
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(kBlock) k_shade(DScene sc, DPass pass, DQueue qin, DQueue qout, DCtrl *c) {
    const uint32_t n = c->n_in;
    ShadeCounters cnt; cnt.bounces = cnt.closest = cnt.shadow = cnt.done = 0;
    // every lane of a wave runs the same number of iterations so the ballots below are wave-complete
    const uint32_t n_round = (n + 63u) & ~63u;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_round; i += gridDim.x * kBlock) {
        const bool active = i < n;
        bool alive = false;
        PathState ps;
        if (active) {
            q_load_for_shade(qin, i, ps);
            alive = shade_path(sc, pass, ps, cnt);
            if (!alive) {
                float *o = pass.sample_rgb + 3ull * ps.sample;
                o[0] = ps.rad.x; o[1] = ps.rad.y; o[2] = ps.rad.z;
                cnt.done++;
            }
        }
        // wave64 stream compaction of the survivors into the next queue
        const unsigned long long mask = __ballot(alive);
        if (mask) {
            const uint32_t lane_off = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            uint32_t base = 0;
            if (lane_off == 0 && alive) base = atomicAdd(&c->n_out, (uint32_t)__popcll(mask));
            base = __shfl(base, __ffsll((long long)mask) - 1, 64);
            if (alive) q_store(qout, base + lane_off, ps);
        }
    }
    uint32_t b = wave_sum(cnt.bounces), cl = wave_sum(cnt.closest), sh = wave_sum(cnt.shadow), dn = wave_sum(cnt.done);
    if ((threadIdx.x & 63) == 0) {
        if (b) atomicAdd(&c->bounce_iterations, (unsigned long long)b);
        if (cl) atomicAdd(&c->rays_closest, (unsigned long long)cl);
        if (sh) atomicAdd(&c->rays_shadow, (unsigned long long)sh);
        if (dn) atomicAdd(&c->samples_done, (unsigned long long)dn);
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
        const float inv = 1.0f / (float)pass.spp;
        rgb[3ull * pixel] = r * inv; rgb[3ull * pixel + 1] = g * inv; rgb[3ull * pixel + 2] = b * inv;
    }
}

// ---------------------------------------------------------------- batched ray queries for the parity tests
struct RayIO { float org[3]; float tnear; float dir[3]; float tfar; };
struct HitIO { float t, u, v; int32_t shape_id, prim_id; };

template <int LDS_NODES, int LDS_PRIMS, int STACK>
__global__ void __launch_bounds__(kBlock) k_trace_rays(DScene sc, const RayIO *rays, long long n, HitIO *hits, unsigned char *occ) {
    __shared__ __attribute__((aligned(16))) DNode lnodes[LDS_NODES];
    __shared__ __attribute__((aligned(16))) DPrim lprims[LDS_PRIMS];
    __shared__ int lstack[STACK * kBlock];
    LdsMem<LDS_NODES, LDS_PRIMS, STACK> mem;
    stage_tree<LDS_NODES, LDS_PRIMS, STACK>(sc, lnodes, lprims, mem.n_lnodes, mem.n_lprims);
    mem.gnodes = sc.nodes; mem.gprims = sc.leaf_prims; mem.spheres = sc.spheres;
    mem.lnodes = lnodes; mem.lprims = lprims; mem.stack = lstack + threadIdx.x;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        RayF ray;
        ray.ox = rays[i].org[0]; ray.oy = rays[i].org[1]; ray.oz = rays[i].org[2];
        ray.dx = rays[i].dir[0]; ray.dy = rays[i].dir[1]; ray.dz = rays[i].dir[2];
        ray.tnear = rays[i].tnear; ray.tfar = rays[i].tfar;
        HitRec h;
        if (occ) { occ[i] = traverse<true>(mem, ray, h) ? 1 : 0; }
        else {
            HitIO o; o.t = 0; o.u = 0; o.v = 0; o.shape_id = -1; o.prim_id = -1;
            if (traverse<false>(mem, ray, h)) {
                const DPrimShade &ps = sc.prims[h.gprim];
                o.t = h.t; o.u = h.u; o.v = h.v; o.shape_id = ps.shape_id; o.prim_id = ps.prim_id;
            }
            hits[i] = o;
        }
    }
}

// ---------------------------------------------------------------- launchers (called from api_device.hip)
// Small scenes: the whole BVH and all primitives sit in LDS (cbox: 38 triangles).  Large scenes: the top of the tree.
constexpr int kSmallNodes = 128, kSmallPrims = 256, kSmallStack = 16;
constexpr int kLargeNodes = 384, kLargePrims = 1, kLargeStack = 40;

bool scene_is_small(int n_nodes, int n_prims, int bvh_depth) { return n_nodes <= kSmallNodes && n_prims <= kSmallPrims && bvh_depth <= kSmallStack; }
int large_stack_depth() { return kLargeStack; }

void launch_prepare(DCtrl *c, hipStream_t s) { hipLaunchKernelGGL(k_prepare, dim3(1), dim3(1), 0, s, c); }
void launch_generate(const DScene &sc, const DPass &pass, const DQueue &q, const DCtrl *c, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_generate, dim3(grid), dim3(kBlock), 0, s, sc, pass, q, c);
}
void launch_extend(const DScene &sc, const DQueue &q, DCtrl *c, bool small, int grid, hipStream_t s) {
    if (small) hipLaunchKernelGGL((k_extend<kSmallNodes, kSmallPrims, kSmallStack>), dim3(grid), dim3(kBlock), 0, s, sc, q, c);
    else hipLaunchKernelGGL((k_extend<kLargeNodes, kLargePrims, kLargeStack>), dim3(grid), dim3(kBlock), 0, s, sc, q, c);
}
void launch_shade(const DScene &sc, const DPass &pass, const DQueue &qin, const DQueue &qout, DCtrl *c, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_shade, dim3(grid), dim3(kBlock), 0, s, sc, pass, qin, qout, c);
}
void launch_resolve(const DPass &pass, uint32_t n_pixels, float *rgb, hipStream_t s) {
    const uint32_t waves_per_block = kBlock / 64;
    const uint32_t grid = (n_pixels + waves_per_block - 1) / waves_per_block;
    if (grid) hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(kBlock), 0, s, pass, n_pixels, rgb);
}
void launch_trace_rays(const DScene &sc, const void *rays, long long n, void *hits, unsigned char *occ, bool small, int grid, hipStream_t s) {
    if (small) hipLaunchKernelGGL((k_trace_rays<kSmallNodes, kSmallPrims, kSmallStack>), dim3(grid), dim3(kBlock), 0, s, sc, (const RayIO *)rays, n, (HitIO *)hits, occ);
    else hipLaunchKernelGGL((k_trace_rays<kLargeNodes, kLargePrims, kLargeStack>), dim3(grid), dim3(kBlock), 0, s, sc, (const RayIO *)rays, n, (HitIO *)hits, occ);
}

} // namespace ljd
