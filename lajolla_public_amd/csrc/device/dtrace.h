// BVH traversal + primitive tests: the device replacement for rtcIntersect1 / rtcOccluded1
// (intersection.cpp:32,83) and the sphere user-geometry callbacks (sphere.inl:40-141).
//
// The primitive tests are the SAME arithmetic, operation for operation, as oracle/lj_oracle.cpp's tri_test /
// sphere_test, compiled with floating-point contraction off (the fused multiply-adds are the ones written out), so a hit record (t, u, v, primitive) is bit-identical
// between the two for the same float ray.  The closest hit is the minimum of (t, global primitive id), which makes
// the result independent of BVH topology and traversal order.
#pragma once
#include "dmath.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace ljd {

struct RayF { float ox, oy, oz, dx, dy, dz, tnear, tfar; };

// a*b - c*d and x*dx + y*dy + z*dz with explicit fused multiply-adds (IEEE fma is exactly specified, so the CPU oracle and
// the GPU agree bit for bit).  Negating (a, c) negates the first exactly and negating (x, y, z) the second, which is what
// keeps the edge tests of two triangles sharing an edge exact mirror images.
#define LJ_TRI_CROSS(a, b, c, d) __builtin_fmaf((a), (b), -((c) * (d)))
#define LJ_TRI_DOT(x, y, z, dx, dy, dz) __builtin_fmaf((z), (dz), __builtin_fmaf((y), (dy), (x) * (dx)))
// Plücker-coordinate edge tests on origin-relative vertices; see oracle/lj_oracle.cpp tri_test for the rationale.
// `tri_test_raw` stops before the barycentric division: it returns t and the unnormalised (U, V, S); u = U * (1 / S),
// v = V * (1 / S).  A traversal keeps (U, V, S) of its closest hit and divides once per ray, not once per accepted hit.
LJ_HD bool tri_test_raw(const RayF &r, float tfar, const float *p0, const float *p1, const float *p2, float &t_out, float &U_out, float &V_out, float &S_out) {
    float ax = p0[0] - r.ox, ay = p0[1] - r.oy, az = p0[2] - r.oz;
    float bx = p1[0] - r.ox, by = p1[1] - r.oy, bz = p1[2] - r.oz;
    float cx = p2[0] - r.ox, cy = p2[1] - r.oy, cz = p2[2] - r.oz;
    float e0x = cx - ax, e0y = cy - ay, e0z = cz - az;
    float e1x = ax - bx, e1y = ay - by, e1z = az - bz;
    float e2x = bx - cx, e2y = by - cy, e2z = bz - cz;
    float s0x = cx + ax, s0y = cy + ay, s0z = cz + az;
    float s1x = ax + bx, s1y = ay + by, s1z = az + bz;
    float s2x = bx + cx, s2y = by + cy, s2z = bz + cz;
    float U = LJ_TRI_DOT(LJ_TRI_CROSS(e0y, s0z, e0z, s0y), LJ_TRI_CROSS(e0z, s0x, e0x, s0z), LJ_TRI_CROSS(e0x, s0y, e0y, s0x), r.dx, r.dy, r.dz);
    float V = LJ_TRI_DOT(LJ_TRI_CROSS(e1y, s1z, e1z, s1y), LJ_TRI_CROSS(e1z, s1x, e1x, s1z), LJ_TRI_CROSS(e1x, s1y, e1y, s1x), r.dx, r.dy, r.dz);
    float W = LJ_TRI_DOT(LJ_TRI_CROSS(e2y, s2z, e2z, s2y), LJ_TRI_CROSS(e2z, s2x, e2x, s2z), LJ_TRI_CROSS(e2x, s2y, e2y, s2x), r.dx, r.dy, r.dz);
    float mn = fminf(fminf(U, V), W), mx = fmaxf(fmaxf(U, V), W);
    if (!(mn >= 0.0f || mx <= 0.0f)) return false;
    // (most tests end above; what is left runs straight through — one exit, no nest of branches around the division)
    float S = (U + V) + W;
    float nx = LJ_TRI_CROSS(e1y, e0z, e1z, e0y), ny = LJ_TRI_CROSS(e1z, e0x, e1x, e0z), nz = LJ_TRI_CROSS(e1x, e0y, e1y, e0x);
    float den = LJ_TRI_DOT(nx, ny, nz, r.dx, r.dy, r.dz);
    float T = LJ_TRI_DOT(nx, ny, nz, ax, ay, az);
    float t = div_ieee(T, den);
    t_out = t; U_out = U; V_out = V; S_out = S;
    return (S != 0.0f) & (den != 0.0f) & (t > r.tnear) & (t <= tfar);
}
LJ_HD bool tri_test(const RayF &r, float tfar, const float *p0, const float *p1, const float *p2, float &t_out, float &u_out, float &v_out) {
    float U, V, S;
    if (!tri_test_raw(r, tfar, p0, p1, p2, t_out, U, V, S)) return false;
    const float rS = div_ieee(1.0f, S);
    u_out = U * rS; v_out = V * rS;
    return true;
}

// sphere.inl:15-38 + 40-84: the reference's callback arithmetic — double maths on the float ray.
LJ_HD bool sphere_test(const RayF &r, const DSphere &s, double &t_out) {
    double ox = r.ox, oy = r.oy, oz = r.oz, dx = r.dx, dy = r.dy, dz = r.dz;
    double vx = ox - s.center[0], vy = oy - s.center[1], vz = oz - s.center[2];
    double A = dx * dx + dy * dy + dz * dz;
    double B = 2 * (dx * vx + dy * vy + dz * vz);
    double C = (vx * vx + vy * vy + vz * vz) - s.radius * s.radius;
    double t0, t1;
    if (A == 0) {
        if (B == 0) return false;
        t0 = t1 = -C / B;
    } else {
        double disc = B * B - 4 * A * C;
        if (disc < 0) return false;
        double rd = sqrt(disc);
        if (B >= 0) { t0 = (-B - rd) / (2 * A); t1 = 2 * C / (-B - rd); }
        else { t0 = 2 * C / (-B + rd); t1 = (-B + rd) / (2 * A); }
    }
    double tn = r.tnear, tf = r.tfar;
    double t = -1;
    if (t0 >= tn && t0 < tf) t = t0;
    if (t1 >= tn && t1 < tf && t < 0) t = t1;
    if (t >= tn && t < tf) { t_out = t; return true; }
    return false;
}

// Slab test of child `k` of a BVH4 node.  The ray's direction signs pick the near and the far plane of every axis, so
// an empty slot (lo = +inf, hi = -inf) can never be entered.  The factor widens the exit distance by 4 ulp: the test
// may accept a box the exact ray misses, never the reverse (boxes are padded by the builder on top of that).
LJ_HD bool box_test4(const DNode4 &nd, int k, const RayF &r, float ix, float iy, float iz, float tfar, float &tentry) {
    const float nx = ix < 0.0f ? nd.hix[k] : nd.lox[k], fx = ix < 0.0f ? nd.lox[k] : nd.hix[k];
    const float ny = iy < 0.0f ? nd.hiy[k] : nd.loy[k], fy = iy < 0.0f ? nd.loy[k] : nd.hiy[k];
    const float nz = iz < 0.0f ? nd.hiz[k] : nd.loz[k], fz = iz < 0.0f ? nd.loz[k] : nd.hiz[k];
    const float t0 = fmaxf(fmaxf((nx - r.ox) * ix, (ny - r.oy) * iy), fmaxf((nz - r.oz) * iz, r.tnear));
    const float t1 = fminf(fminf((fx - r.ox) * ix, (fy - r.oy) * iy), fminf((fz - r.oz) * iz, tfar));
    tentry = t0;
    return t0 <= t1 * 1.0000005f;
}

struct HitRec { float t, u, v; int32_t gprim; };

// Closest-hit / any-hit update for one primitive of a leaf; returns true when an any-hit query is finished.
template <bool ANY_HIT, class Mem>
LJ_HD bool leaf_prim_test(Mem &mem, const RayF &ray, const DPrim &p, HitRec &best) {
    float t, u = 0.0f, v = 0.0f;
    if (p.kind == 0) {
        if (!tri_test(ray, best.t, p.v0, p.v1, p.v2, t, u, v)) return false;
    } else {
        double td;
        if (!sphere_test(ray, mem.sphere(p.sphere_slot), td)) return false;
        t = (float)td;
    }
    if (ANY_HIT) { best.t = t; best.gprim = p.gprim; return true; }
    if (t < best.t || (t == best.t && (best.gprim < 0 || p.gprim < best.gprim))) { best.t = t; best.u = u; best.v = v; best.gprim = p.gprim; }
    return false;
}

// Reference form of the traversal the extend kernel performs (kernels.hip keeps a wave-synchronous version of the same
// steps): hit children are visited nearest first, the others wait on the stack.
// Mem provides: DNode4 node(int i); DPrim prim(int i); const DSphere& sphere(int slot);
//               void push(int sp, int v); int pop(int sp);   (per-lane stack storage, 3 * bvh depth entries)
template <bool ANY_HIT, class Mem>
LJ_HD bool traverse(Mem &mem, const RayF &ray, HitRec &best) {
    best.t = ray.tfar; best.u = 0.0f; best.v = 0.0f; best.gprim = -1;
    const float ix = 1.0f / ray.dx, iy = 1.0f / ray.dy, iz = 1.0f / ray.dz;
    int sp = 0;
    int cur = 0;
    for (;;) {
        if (cur >= 0) {
            const DNode4 nd = mem.node(cur);
            float td[4]; int cd[4]; int nh = 0;
            for (int k = 0; k < 4; k++) {
                float te;
                if (!box_test4(nd, k, ray, ix, iy, iz, best.t, te)) continue;
                int pos = nh++;
                while (pos > 0 && td[pos - 1] > te) { td[pos] = td[pos - 1]; cd[pos] = cd[pos - 1]; pos--; }
                td[pos] = te; cd[pos] = nd.child[k];
            }
            if (nh == 0) {
                if (sp == 0) break;
                sp--; cur = mem.pop(sp);
            } else {
                for (int k = nh - 1; k >= 1; k--) { mem.push(sp, cd[k]); sp++; }
                cur = cd[0];
            }
        } else {
            const int first = (~cur) >> 3, count = ((~cur) & 7) + 1;
            for (int k = 0; k < count; k++)
                if (leaf_prim_test<ANY_HIT>(mem, ray, mem.prim(first + k), best)) return true;
            if (sp == 0) break;
            sp--; cur = mem.pop(sp);
        }
    }
    return best.gprim >= 0;
}

// ---- BVH8 (DNode8): slab test of all eight children on the node's 8-bit grid, and the octant-ordered traversal.
// plane = p + q * 2^(e-127); t = (plane - o) * i = q * (2^(e-127) * i) + (p * i - o * i): one conversion and one fma per plane.
struct Ray8 {
    float ix, iy, iz, oix, oiy, oiz;   // 1 / d and o / d
    uint32_t oct;                      // bit k: the ray runs towards the low side of axis k
};
LJ_HD float grid_step(uint32_t biased_exponent) {   // 2^(e - 127), e in 1 .. 254
    union { uint32_t u; float f; } c; c.u = biased_exponent << 23; return c.f;
}
LJ_HD float byte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }   // v_cvt_f32_ubyte<k> on the device
// w[0..19]: the node as twenty dwords.  Returns the slots (bit s = slot s) whose quantised box the ray segment [tnear, tfar] enters;
// like box_test4 the exit distance is widened by 4 ulp, and an axis the ray does not move along (i = inf: every plane reads nan) drops out.
LJ_HD uint32_t node8_hits(const uint32_t *w, const Ray8 &r, float tnear, float tfar) {
    union { uint32_t u; float f; } px, py, pz; px.u = w[0]; py.u = w[1]; pz.u = w[2];
    const float ax = grid_step(w[3] & 0xffu) * r.ix, ay = grid_step((w[3] >> 8) & 0xffu) * r.iy, az = grid_step((w[3] >> 16) & 0xffu) * r.iz;
    const float bx = __builtin_fmaf(px.f, r.ix, -r.oix), by = __builtin_fmaf(py.f, r.iy, -r.oiy), bz = __builtin_fmaf(pz.f, r.iz, -r.oiz);
    // dwords 8..19: qlo_x[0-3] qlo_x[4-7] qlo_y.. qlo_y.. | qlo_z qlo_z qhi_x qhi_x | qhi_y qhi_y qhi_z qhi_z
    const bool nx = (r.oct & 1u) != 0u, ny = (r.oct & 2u) != 0u, nz = (r.oct & 4u) != 0u;
    // Slots 7 .. 0, each shifting the sign of (entry - 1.0000005 exit) into the mask from below (one fma and one v_alignbit per slot instead of
    // multiply, compare, select, or): negative = the segment enters the box.  Against `entry <= round(exit * 1.0000005)` the decision can
    // differ only within one rounding of the 4-ulp allowance — the test stays conservative, and which boxes are entered never changes a hit.
    uint32_t hits = 0u;
    for (int h = 1; h >= 0; h--) {
        const uint32_t lox = w[8 + h], loy = w[10 + h], loz = w[12 + h], hix = w[14 + h], hiy = w[16 + h], hiz = w[18 + h];
        const uint32_t nearx = nx ? hix : lox, farx = nx ? lox : hix;
        const uint32_t neary = ny ? hiy : loy, fary = ny ? loy : hiy;
        const uint32_t nearz = nz ? hiz : loz, farz = nz ? loz : hiz;
        for (int k = 3; k >= 0; k--) {
            const float t0 = fmaxf(fmaxf(__builtin_fmaf(byte_f(nearx, k), ax, bx), __builtin_fmaf(byte_f(neary, k), ay, by)), fmaxf(__builtin_fmaf(byte_f(nearz, k), az, bz), tnear));
            const float t1 = fminf(fminf(__builtin_fmaf(byte_f(farx, k), ax, bx), __builtin_fmaf(byte_f(fary, k), ay, by)), fminf(__builtin_fmaf(byte_f(farz, k), az, bz), tfar));
            union { float f; uint32_t u; } d; d.f = __builtin_fmaf(t1, -1.0000005f, t0);
            hits = (hits << 1) | (d.u >> 31);
        }
    }
    return hits;
}
// bit s of m moves to bit (s ^ oct): after this the lowest set bit is the slot the ray meets first
LJ_HD uint32_t perm8(uint32_t m, uint32_t oct) {
    uint32_t t = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u); m = (oct & 1u) ? t : m;
    t = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u); m = (oct & 2u) ? t : m;
    t = ((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu); m = (oct & 4u) ? t : m;
    return m;
}
LJ_HD Ray8 ray8_setup(const RayF &ray) {
    Ray8 r;
    r.ix = recip_fast(ray.dx); r.iy = recip_fast(ray.dy); r.iz = recip_fast(ray.dz);
    r.oix = ray.ox * r.ix; r.oiy = ray.oy * r.iy; r.oiz = ray.oz * r.iz;
    r.oct = (r.ix < 0.0f ? 1u : 0u) | (r.iy < 0.0f ? 2u : 0u) | (r.iz < 0.0f ? 4u : 0u);
    return r;
}
LJ_HD int ctz32(uint32_t v) { return __builtin_ctz(v); }
LJ_HD int popc32(uint32_t v) { return __builtin_popcount(v); }

// Reference form of the BVH8 traversal (kernels.hip keeps a wave-synchronous version of the same steps).  A node group is
// (child_base, remaining inner hits in octant order | imask << 8): one stack entry per visited node, however many of its children were hit.
// Mem provides: const uint32_t *node8(int i); DPrim prim(int i); const DSphere& sphere(int slot);
//               void push8(int sp, uint32_t base, uint32_t bits); void pop8(int sp, uint32_t &base, uint32_t &bits);
template <bool ANY_HIT, class Mem>
LJ_HD bool traverse8(Mem &mem, const RayF &ray, HitRec &best) {
    best.t = ray.tfar; best.u = 0.0f; best.v = 0.0f; best.gprim = -1;
    const Ray8 r8 = ray8_setup(ray);
    uint32_t gbase = 0u, gbits = 1u;   // the root as a group of one: imask 0, so the child picked is node `gbase`
    int sp = 0;
    for (;;) {
        if ((gbits & 0xffu) == 0u) {
            if (sp == 0) break;
            sp--; mem.pop8(sp, gbase, gbits);
        }
        const uint32_t k = (uint32_t)ctz32(gbits & 0xffu), s = k ^ r8.oct, imask = (gbits >> 8) & 0xffu;
        gbits &= gbits - 1u;
        const uint32_t node = gbase + (uint32_t)popc32(imask & ((1u << s) - 1u));
        if (gbits & 0xffu) { mem.push8(sp, gbase, gbits); sp++; }
        const uint32_t *w = mem.node8((int)node);
        const uint32_t hits = node8_hits(w, r8, ray.tnear, best.t);
        const uint32_t nmask = w[3] >> 24, inner = hits & nmask;
        uint32_t leaf = hits & ~nmask;
        while (leaf) {
            const int slot = ctz32(leaf); leaf &= leaf - 1u;
            const uint32_t m = (w[6 + (slot >> 2)] >> (8 * (slot & 3))) & 0xffu;
            if (!(m & 0x80u)) continue;   // (an empty slot can pass the widened test of a degenerate node)
            const int first = (int)(w[5] + (m & 31u)), count = (int)((m >> 5) & 3u) + 1;
            for (int i = 0; i < count; i++)
                if (leaf_prim_test<ANY_HIT>(mem, ray, mem.prim(first + i), best)) return true;
        }
        gbase = w[4]; gbits = inner ? (perm8(inner, r8.oct) | (nmask << 8)) : 0u;
    }
    return best.gprim >= 0;
}

} // namespace ljd

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
