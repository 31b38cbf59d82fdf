// Per-lane shading logic of the wavefront integrator, in float: camera rays, shading info, textures, lights,
// BSDFs and the bounce step of path_tracing() re-cut into "everything between two ray casts".
// Each function cites the reference code it computes the same thing as.  Compiled for gfx950 by hipcc and, for
// CPU-side debugging of the very same source, by g++ in tests/twin (never shipped, never a fallback).
#pragma once
#include "dmath.h"
#include "dconfig.h"

namespace ljd {

// ------------------------------------------------------------------ textures (texture.h:123-154, mipmap.h:52-89)
// Compile-time description of what a scene can contain.  The shade kernel is instantiated for a few feature sets
// (kernels.hip) and the scene upload picks the smallest one that covers the scene: code for absent Material / Texture /
// Light alternatives is not generated at all, which cuts registers, spills and instruction-cache pressure.
// KINDS: bit k set = Material alternative k may occur.  TEXTURED: image / checkerboard textures may occur.
// ENVMAP: an environment map light may occur.  SPHERE_LIGHTS: spherical area lights may occur.
template <uint32_t KINDS, bool TEXTURED, bool ENVMAP, bool SPHERE_LIGHTS>
struct ShadeFeat {
    static constexpr uint32_t kinds = KINDS;
    static constexpr bool textured = TEXTURED, envmap = ENVMAP, sphere_lights = SPHERE_LIGHTS;
    static constexpr bool kind(int k) { return ((KINDS >> k) & 1u) != 0u; }
};
using FeatAll = ShadeFeat<0x1ffu, true, true, true>;
// The feature sets device code is compiled for, smallest first; variant v of a scene = the first one that
// covers its Material / Texture / Light alternatives.  kinds: bit k = Material alternative k (material.h:102-110).
using FeatLambert = ShadeFeat<0x001u, false, false, false>;      // constant-colour diffuse surfaces, mesh lights (cbox)
using FeatPlastic = ShadeFeat<0x003u, false, false, true>;       // diffuse + roughplastic in constant colours, mesh and sphere lights (veach_mi)
using FeatLambertTex = ShadeFeat<0x001u, true, false, true>;     // + image / checker textures, sphere lights (sponza)
using FeatClassic = ShadeFeat<0x007u, true, true, true>;         // diffuse, roughplastic, roughdielectric + everything else
using FeatDisney = ShadeFeat<0x101u, true, true, false>;         // diffuse + the Disney principled BSDF, textures, environment map (disney_bsdf.xml)
// (kNumShadeVariants = 6 and kShadeVariantAll, the index of the one that covers everything, live in dconfig.h)
// calls fn(Feat{}) for variant v
template <class Fn> inline void with_shade_variant(int v, Fn &&fn) {
    switch (v) {
        case 0: fn(FeatLambert{}); break;
        case 1: fn(FeatPlastic{}); break;
        case 2: fn(FeatLambertTex{}); break;
        case 3: fn(FeatClassic{}); break;
        case 4: fn(FeatDisney{}); break;
        default: fn(FeatAll{}); break;
    }
}
inline bool variant_covers(int v, uint32_t kinds, bool textured, bool envmap, bool sphere_lights) {
    bool ok = false;
    with_shade_variant(v, [&](auto ft) {
        using Ft = decltype(ft);
        ok = (kinds & ~Ft::kinds) == 0u && (Ft::textured || !textured) && (Ft::envmap || !envmap) && (Ft::sphere_lights || !sphere_lights);
    });
    return ok;
}

struct alignas(16) Texel4 { float r, g, b, pad; };
LJ_HD f3 texel(const DScene &sc, const DImage &img, int level, int x, int y) {
    const DMipLevel lv = img.lv[level];
    if (img.channels >= 3) {   // RGB0 quads, 16-byte aligned (flatten.cpp build_mips): one 16-byte load
        const Texel4 t = *reinterpret_cast<const Texel4 *>(sc.texels + lv.offset + ((int64_t)y * lv.w + x) * 4);
        return mk3(t.r, t.g, t.b);
    }
    const float v = sc.texels[lv.offset + ((int64_t)y * lv.w + x)];
    return mk3(v, v, v);
}
// Texture coordinates are carried in double: tiled uvs reach tens of units and a 1000-texel image needs ~1e-8 relative
// precision in the fractional part for the bilinear weights to agree with the reference's double arithmetic; MI355X
// runs fp64 at half the fp32 rate, and only the addressing is double — texel blending stays float.
// modulo(i, n) (lajolla.h:55-61) for the index of a texel.  The callers hand in u in [0, 1], so (int)(u * n - 0.5) already lies in
// [0, n - 1] and the integer division behind `%` (~30 instructions, eight of them per trilinear lookup) is never needed: it stays as
// the out-of-range path, which no lane takes.
LJ_HD int wrap_index(int i, int n) { return ((unsigned)i < (unsigned)n) ? i : moduloi(i, n); }
LJ_HD int wrap_next(int i, int n) { return (i + 1 == n) ? 0 : wrap_index(i + 1, n); }
LJ_HD f3 mip_lookup_level(const DScene &sc, const DImage &img, double u, double v, int level) {
    const int W = img.lv[level].w, H = img.lv[level].h;
    u = u * W - 0.5; v = v * H - 0.5;
    int ufi = wrap_index((int)u, W), vfi = wrap_index((int)v, H);
    int uci = wrap_next(ufi, W), vci = wrap_next(vfi, H);
    float uo = (float)(u - ufi), vo = (float)(v - vfi);
    f3 ff = texel(sc, img, level, ufi, vfi), fc = texel(sc, img, level, ufi, vci);
    f3 cf = texel(sc, img, level, uci, vfi), cc = texel(sc, img, level, uci, vci);
    return ff * ((1 - uo) * (1 - vo)) + fc * ((1 - uo) * vo) + cf * (uo * (1 - vo)) + cc * (uo * vo);
}
LJ_HD f3 mip_lookup(const DScene &sc, const DImage &img, double u, double v, float level) {
    if (level <= 0.0f) return mip_lookup_level(sc, img, u, v, 0);
    if (level < (float)(img.levels - 1)) {
        int fl = (int)floorf(level); fl = fl < 0 ? 0 : (fl > img.levels - 1 ? img.levels - 1 : fl);
        int cl = fl + 1 > img.levels - 1 ? img.levels - 1 : fl + 1;
        float lo = level - fl;
        return mip_lookup_level(sc, img, u, v, fl) * (1 - lo) + mip_lookup_level(sc, img, u, v, cl) * lo;
    }
    return mip_lookup_level(sc, img, u, v, img.levels - 1);
}
LJ_HD double modulod(double a, double b) { double r = fmod(a, b); return (r < 0.0) ? r + b : r; }
// modulo(a, 1) (lajolla.h:63-66: fmod, then + 1 if negative) without the fmod: a - floor(a) is the same number, rounded the same way
// (a - trunc(a) is exact, and the reference's r + 1 and this a - floor(a) round the same real), at two instructions instead of fmod's loop.
LJ_HD double modulo1(double a) { return a - floor(a); }
template <class Ft = FeatAll>
LJ_HD f3 eval_texture(const DScene &sc, const DTexture &t, bool spectrum, double u, double v, float footprint) {
    if (!Ft::textured || t.kind == 0) return ld3(t.value);
    double lu = modulo1(u * (double)t.uscale + (double)t.uoffset), lv = modulo1(v * (double)t.vscale + (double)t.voffset);
    if (t.kind == 1) {
        const DImage &img = spectrum ? sc.images3[t.texture_id] : sc.images1[t.texture_id];
        float scaled = (float)(img.lv[0].w > img.lv[0].h ? img.lv[0].w : img.lv[0].h) * fmaxf(t.uscale, t.vscale) * footprint;
        float level = log2f(fmaxf(scaled, 1e-8f));
        return mip_lookup(sc, img, lu, lv, level);
    }
    int x = 2 * ((int)(lu * 2) & 1) - 1, y = 2 * ((int)(lv * 2) & 1) - 1;   // modulo(int(2 u), 2) of a non-negative u
    return (x * y == 1) ? ld3(t.value) : ld3(t.color1);
}

// ------------------------------------------------------------------ camera (camera.cpp:23-47, filters/*.inl)
LJ_HD void filter_sample(int kind, float param, float r0, float r1, float &ox, float &oy) {
    if (kind == 0) { ox = (2.0f * r0 - 1.0f) * (param / 2); oy = (2.0f * r1 - 1.0f) * (param / 2); }
    else if (kind == 1) {
        float h = param / 2;
        ox = r0 < 0.5f ? h * (sqrtf(2 * r0) - 1) : h * (1 - sqrtf(1 - 2 * (r0 - 0.5f)));
        oy = r1 < 0.5f ? h * (sqrtf(2 * r1) - 1) : h * (1 - sqrtf(1 - 2 * (r1 - 0.5f)));
    } else {
        float r = param * sqrtf(-2.0f * logf(fmaxf(r0, 1e-8f)));
        float sn, cs; sincos_2pi(r1, sn, cs);
        ox = r * cs; oy = r * sn;
    }
}
LJ_HD f3 xform_point16(const float *m, f3 p) {
    float x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    float inv = 1.0f / w;
    return mk3(x * inv, y * inv, z * inv);
}
LJ_HD f3 xform_vector16(const float *m, f3 v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
LJ_HD f3 xform_vector9(const float *m, f3 v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
// Primary ray for pixel (x, y) with jitter (jx, jy) in [0,1).  The reference recovers the sub-pixel offset as
// frac(((x + jx) / w) * w) (camera.cpp:26-31); in exact arithmetic that is jx, and using jx directly avoids a float
// rounding that could move a sample into the neighbouring pixel.
LJ_HD f3 camera_primary_dir(const DCamera &cam, int x, int y, float jx, float jy) {
    float ox, oy;
    filter_sample(cam.filter_kind, cam.filter_param, jx, jy, ox, oy);
    float rx = ((float)x + 0.5f + ox) / (float)cam.width, ry = ((float)y + 0.5f + oy) / (float)cam.height;
    f3 pt = xform_point16(cam.sample_to_cam, mk3(rx, ry, 0.0f));
    return normalize(xform_vector16(cam.cam_to_world, normalize(pt)));
}

// ------------------------------------------------------------------ table distributions (table_dist.cpp)
// std::upper_bound(cdf, cdf + n + 1, u) - 1, clamped to [0, n-1]
LJ_HD int sample_cdf(const float *cdf, int n, float u) {
    int lo = 0, hi = n + 1;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cdf[mid] > u) hi = mid; else lo = mid + 1; }
    int off = lo - 1;
    return off < 0 ? 0 : (off > n - 1 ? n - 1 : off);
}

// The same search with a guide table: entry b of `guide` (n entries, one per 1/n of the unit interval, built on the host: flatten.cpp
// make_cdf_guide) brackets the answer for every u whose bin is b — low 16 bits: first candidate, high 16 bits: last — so the bisection
// runs over a handful of entries instead of all n + 1: an environment map's two searches are 21 dependent loads otherwise, the longest
// latency chain of a shade step.  Same comparisons on the same floats: the same index as sample_cdf, always.
LJ_HD int sample_cdf_guided(const float *cdf, const float *guide, int n, float u) {
    int b = (int)(u * (float)n);
    b = b < 0 ? 0 : (b > n - 1 ? n - 1 : b);
    union { float f; uint32_t u; } g; g.f = guide[b];
    int lo = (int)(g.u & 0xffffu), hi = (int)(g.u >> 16);
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cdf[mid] > u) hi = mid; else lo = mid + 1; }
    int off = lo - 1;
    return off < 0 ? 0 : (off > n - 1 ? n - 1 : off);
}
// ... returning cdf[off] and cdf[off + 1] as well (what inverting the piecewise-linear cdf needs next).  A bracket of at most four entries —
// the usual case — is fetched at once with its two neighbours: six independent loads and a count instead of bisection steps that each
// wait for the one before, and the two values come out of the same six.
LJ_HD int sample_cdf_guided(const float *cdf, const float *guide, int n, float u, float &c0, float &c1) {
    int b = (int)(u * (float)n);
    b = b < 0 ? 0 : (b > n - 1 ? n - 1 : b);
    union { float f; uint32_t u; } g; g.f = guide[b];
    int lo = (int)(g.u & 0xffffu), hi = (int)(g.u >> 16);
    if (hi - lo <= 4) {
        float v[6];
        for (int k = 0; k < 6; k++) { int i = lo - 1 + k; i = i < 0 ? 0 : (i > n ? n : i); v[k] = cdf[i]; }
        int r = lo;   // the first entry greater than u = lo + the entries of [lo, hi) that are not (the cdf is non-decreasing)
        for (int k = 1; k <= 4; k++) r += (lo - 1 + k < hi && v[k] <= u) ? 1 : 0;
        const int off = r - 1;
        if (off >= 0 && off <= n - 1) {
            const int k0 = off - (lo - 1);   // 0 .. 4
            c0 = k0 == 0 ? v[0] : (k0 == 1 ? v[1] : (k0 == 2 ? v[2] : (k0 == 3 ? v[3] : v[4])));
            c1 = k0 == 0 ? v[1] : (k0 == 1 ? v[2] : (k0 == 2 ? v[3] : (k0 == 3 ? v[4] : v[5])));
            return off;
        }
        lo = r;   // (clamped below)
    } else {
        while (lo < hi) { int mid = (lo + hi) >> 1; if (cdf[mid] > u) hi = mid; else lo = mid + 1; }
    }
    int off = lo - 1;
    off = off < 0 ? 0 : (off > n - 1 ? n - 1 : off);
    c0 = cdf[off]; c1 = cdf[off + 1];
    return off;
}

// ------------------------------------------------------------------ path vertex (intersection.cpp:38-62)
struct DVertex {
    f3 position, gn;
    Frame3 frame;
    double u, v;             // texture uv (double: see mip_lookup_level)
    float uv_screen_size;
    int32_t material_id, light_id, gprim;
    bool is_sphere;
};

LJ_HD DVertex build_vertex(const DScene &sc, f3 org, f3 dir, float t, float bu, float bv, int gprim, float ray_spread) {
    DVertex vx;
    const DPrimShade &ps = sc.prims[gprim];
    // one rounding per coordinate (the hit point feeds the [eps, ...) self-intersection guard of the next rays)
    vx.position = mk3(fmaf(dir.x, t, org.x), fmaf(dir.y, t, org.y), fmaf(dir.z, t, org.z));
    vx.material_id = ps.material_id; vx.light_id = ps.light_id; vx.gprim = gprim;
    vx.is_sphere = (ps.flags & 1) != 0;
    float inv_uv_size;
    if (!vx.is_sphere) {  // triangle_mesh.inl:65-157
        float b0 = 1.0f - bu - bv;
        {   // (1 - s - t) uv0 + s uv1 + t uv2 with s, t the float barycentrics widened, as the reference does (triangle_mesh.inl:81-83)
            const double s_ = bu, t_ = bv, b0d = 1.0 - s_ - t_;
            vx.u = b0d * ps.uv0[0] + s_ * ps.uv1[0] + t_ * ps.uv2[0];
            vx.v = b0d * ps.uv0[1] + s_ * ps.uv1[1] + t_ * ps.uv2[1];
        }
        f3 gn = ld3(ps.gn), sn = gn;
        if (ps.flags & 2) sn = normalize(ld3(ps.n0) * b0 + ld3(ps.n1) * bu + ld3(ps.n2) * bv);
        f3 dpdu = ld3(ps.dpdu);
        f3 tangent = normalize(dpdu - sn * dot(sn, dpdu));
        f3 bitangent = normalize(cross(sn, tangent));
        vx.frame.x = tangent; vx.frame.y = bitangent; vx.frame.n = sn;
        vx.gn = dot(gn, sn) < 0.0f ? -gn : gn;
        inv_uv_size = ps.inv_uv_size;
    } else {  // sphere.inl:85-99 (Ng, st) and 235-260 (shading info; st consumed as radians, as written)
        f3 c = ld3(ps.n0); float r = ps.n1[0];
        f3 N = vx.position - c;
        f3 gn = normalize(N);
        f3 cart = N / r;
        float su = atan2f(cart.z, cart.x) * kInvTwoPi, sv = acosf(clampf(cart.y, -1.0f, 1.0f)) * kInvPi;
        f3 dpdu = mk3(-r * sinf(su) * sinf(sv), r * cosf(su) * sinf(sv), 0.0f);
        f3 dpdv = mk3(r * cosf(su) * cosf(sv), r * sinf(su) * cosf(sv), -r * sinf(sv));
        f3 tangent = normalize(dpdu - gn * dot(gn, dpdu));
        vx.frame.x = tangent; vx.frame.y = normalize(cross(gn, tangent)); vx.frame.n = gn;
        vx.gn = gn; vx.u = su; vx.v = sv;
        inv_uv_size = (length(dpdu) + length(dpdv)) * 0.5f;
    }
    float ray_radius = ray_spread * length(org - vx.position);  // transfer(): radius is always 0 in path_tracing (ray.h:35-42)
    vx.uv_screen_size = ray_radius / inv_uv_size;
    return vx;
}

// ------------------------------------------------------------------ auxiliary buffers (render.cpp:12-69 aux_render)
// The value of one pixel for the Depth / ShadingNormal / MeanCurvature / RayDifferential / MipmapLevel "integrators":
// one primary ray through the pixel centre, no random numbers.  `integrator` uses the LJ_INTEGRATOR_* numbering (0..4).
LJ_HD f3 aux_value(const DScene &sc, int integrator, f3 org, f3 dir, float t, float bu, float bv, int gprim) {
    if (gprim < 0) return mk3(0, 0, 0);
    const DVertex vx = build_vertex(sc, org, dir, t, bu, bv, gprim, sc.init_spread);
    const DPrimShade &ps = sc.prims[gprim];
    if (integrator == 0) { float d = length(vx.position - org); return mk3(d, d, d); }
    if (integrator == 1) return vx.frame.n;
    if (integrator == 2) {  // triangle_mesh.inl:128-146, sphere.inl:258
        float k = 0.0f;
        if (vx.is_sphere) k = 1.0f / ps.n1[0];
        else if (ps.flags & 2) {
            // the uv determinant cancels badly on nearly degenerate uv maps: double, as the reference has it
            const double dus = (double)ps.uv2[0] - ps.uv0[0], dvs = (double)ps.uv2[1] - ps.uv0[1], dut = (double)ps.uv2[0] - ps.uv1[0], dvt = (double)ps.uv2[1] - ps.uv1[1];
            const double det = dus * dvt - dut * dvs;
            const float dsdu = (float)(dvt / det), dtdu = (float)(-dvs / det), dsdv = (float)(dut / det), dtdv = (float)(-dus / det);
            const f3 dnds = ld3(ps.n2) - ld3(ps.n0), dndt = ld3(ps.n2) - ld3(ps.n1);
            const f3 dndu = dnds * dsdu + dndt * dtdu, dndv = dnds * dsdv + dndt * dtdv;
            k = (dot(dndu, vx.frame.x) + dot(dndv, vx.frame.y)) * 0.5f;
        }
        return mk3(k, k, k);
    }
    if (integrator == 3) return mk3(0.0f, sc.init_spread, 0.0f);   // init_ray_differential (ray.h:35-37): radius 0
    // MipmapLevel: only for an image texture in the material's get_texture() slot (materials/*.inl get_texture_op)
    const DMaterial &m = sc.materials[vx.material_id];
    if (m.kind == 6 || m.tex[0].kind != 1) return mk3(0, 0, 0);
    const DTexture &tx = m.tex[0];
    const DImage &img = sc.images3[tx.texture_id];
    const float scaled = (float)(img.lv[0].w > img.lv[0].h ? img.lv[0].w : img.lv[0].h) * fmaxf(tx.uscale, tx.vscale) * vx.uv_screen_size;
    const float level = log2f(fmaxf(scaled, 1e-8f));
    return mk3(level, level, level);
}

// ------------------------------------------------------------------ lights (lights/*.inl, shapes/*.inl sampling)
// `dpos`: the sampled point in double.  For a sphere light the direction and distance to it must come from this one: a
// float point sits up to 1e-6 off the sphere, and from a grazing angle a shadow ray then meets the light's own surface
// some sqrt(2 r 1e-6) before the point — outside the (1 - eps) margin — which shadowed 12 % of the samples of a small,
// distant sphere light (disney_bsdf_test/simple_sphere.xml).  Triangle lights: the float point, widened.
struct LightSample { f3 position, normal; double dpos[3]; };

LJ_HD float sphere_one_minus_cos_max(float r, float dist_sq) {
    // 1 - sqrt(1 - r^2/d^2) without cancellation: s / (1 + sqrt(1 - s))
    float s = r * r / dist_sq;
    return s / (1.0f + sqrtf(fmaxf(0.0f, 1.0f - s)));
}

template <class Ft = FeatAll>
LJ_HD LightSample sample_point_on_light(const DScene &sc, const DLight &L, f3 ref, float u0, float u1, float w) {
    LightSample ls;
    if (!Ft::envmap || L.kind == 0) {
        if (!Ft::sphere_lights || !L.is_sphere) {  // triangle_mesh.inl:24-38
            int tri = sample_cdf(sc.light_tri_cdf + L.cdf_first, L.tri_count, w);
            const DLightTri &T = sc.light_tris[L.tri_first + tri];
            float a = sqrtf(clampf(u0, 0.0f, 1.0f));
            float b1 = 1.0f - a, b2 = a * u1;
            ls.position = ld3(T.v0) + ld3(T.e1) * b1 + ld3(T.e2) * b2;
            ls.normal = ld3(T.n);
            ls.dpos[0] = ls.position.x; ls.dpos[1] = ls.position.y; ls.dpos[2] = ls.position.z;
        } else {  // sphere.inl:156-204, in the reference's double arithmetic (see LightSample)
            const double cx = L.center[0], cy = L.center[1], cz = L.center[2], r = L.radius;
            const double vx = cx - (double)ref.x, vy = cy - (double)ref.y, vz = cz - (double)ref.z;
            const double dist_sq = vx * vx + vy * vy + vz * vz;
            double nx, ny, nz;
            if (dist_sq < r * r) {
                const double z = 1.0 - 2.0 * (double)u0;
                const double r_ = sqrt(fmax(0.0, 1.0 - z * z));
                float sn, cs; sincos_2pi(u1, sn, cs);   // (the azimuth does not need double: see below)
                nx = r_ * (double)cs; ny = r_ * (double)sn; nz = z;
            } else {
                const double dc = sqrt(dist_sq);
                const double dx = vx / dc, dy = vy / dc, dz = vz / dc;   // dir_to_center
                // coordinate_system (frame.h:11-22)
                double ax, ay, az, bx, by, bz;
                if (dz < -1.0 + 1e-6) { ax = 0; ay = -1; az = 0; bx = -1; by = 0; bz = 0; }
                else {
                    const double a = 1.0 / (1.0 + dz), b = -dx * dy * a;
                    ax = 1.0 - dx * dx * a; ay = b; az = -dx;
                    bx = b; by = 1.0 - dy * dy * a; bz = -dy;
                }
                const double sin_max_sq = r * r / dist_sq;
                const double cos_max = sqrt(fmax(0.0, 1.0 - sin_max_sq));
                const double cos_el = (1.0 - (double)u0) + (double)u0 * cos_max;
                const double sin_el = sqrt(fmax(0.0, 1.0 - cos_el * cos_el));
                // What needs double here is the point's distance from the centre (the elevation chain below cancels for a small or
                // distant light); the azimuth only moves the point ALONG the sphere, so its sin / cos come from the float samplers'
                // one-instruction sincos (a double sin + cos pair is ~200 instructions of every path-step of a scene lit by a sphere).
                float sn, cs; sincos_2pi(u1, sn, cs);
                const double ds = dc * cos_el - sqrt(fmax(0.0, r * r - dc * dc * sin_el * sin_el));
                const double cos_alpha = (dc * dc + r * r - ds * ds) / (2.0 * dc * r);
                const double sin_alpha = sqrt(fmax(0.0, 1.0 - cos_alpha * cos_alpha));
                const double lx = sin_alpha * (double)cs, ly = sin_alpha * (double)sn, lz = cos_alpha;
                nx = -(ax * lx + bx * ly + dx * lz); ny = -(ay * lx + by * ly + dy * lz); nz = -(az * lx + bz * ly + dz * lz);
            }
            ls.dpos[0] = cx + r * nx; ls.dpos[1] = cy + r * ny; ls.dpos[2] = cz + r * nz;
            ls.position = mk3((float)ls.dpos[0], (float)ls.dpos[1], (float)ls.dpos[2]);
            ls.normal = mk3((float)nx, (float)ny, (float)nz);
        }
    } else {  // envmap.inl:7-20 with table_dist.cpp:116-139
        const float *cm = sc.env_marg + (L.env_cdf_marg - sc.env_marg_first);
        float c0, c1;
        int yo = sample_cdf_guided(cm, sc.env_marg + (L.env_guide_marg - sc.env_marg_first), L.env_h, u1, c0, c1);
        float dy = u1 - c0;
        if (c1 - c0 > 0.0f) dy /= (c1 - c0);
        const float *cdf = sc.env_tables + L.env_cdf_rows + (int64_t)yo * (L.env_w + 1);
        int xo = sample_cdf_guided(cdf, sc.env_tables + L.env_guide_rows + (int64_t)yo * L.env_w, L.env_w, u0, c0, c1);
        float dx = u0 - c0;
        if (c1 - c0 > 0.0f) dx /= (c1 - c0);
        float az = ((xo + dx) / L.env_w) * kTwoPi, el = ((yo + dy) / L.env_h) * kPi;
        f3 local = mk3(sinf(az) * sinf(el), cosf(el), -cosf(az) * sinf(el));
        ls.position = mk3(0, 0, 0); ls.normal = -xform_vector9(L.to_world, local);
        ls.dpos[0] = ls.dpos[1] = ls.dpos[2] = 0.0;
    }
    return ls;
}

LJ_HD void envmap_dir_to_uv(f3 local, float &u, float &v) {
    u = atan2f(local.x, -local.z) * kInvTwoPi; v = acosf(clampf(local.y, -1.0f, 1.0f)) * kInvPi;
    if (u < 0.0f) u += 1.0f;
}

template <class Ft = FeatAll>
LJ_HD float pdf_point_on_light(const DScene &sc, const DLight &L, f3 pos, f3 nrm, f3 ref) {
    if (!Ft::envmap || L.kind == 0) {
        if (!Ft::sphere_lights || !L.is_sphere) return 1.0f / L.total_area;  // triangle_mesh.inl:44-46
        f3 center = ld3(L.center); float r = L.radius;  // sphere.inl:210-230
        f3 dc_vec = ref - center;
        float dist_sq = dot(dc_vec, dc_vec);
        if (dist_sq < r * r) return 1.0f / (4.0f * kPi * r * r);
        float pdf_solid_angle = 1.0f / (kTwoPi * sphere_one_minus_cos_max(r, dist_sq));
        f3 d = pos - ref;
        float d2 = dot(d, d);
        f3 dir = normalize(d);
        return pdf_solid_angle * fabsf(dot(nrm, dir)) / d2;
    }
    f3 local = xform_vector9(L.to_local, -nrm);  // envmap.inl:22-42, table_dist.cpp:141-151
    float u, v; envmap_dir_to_uv(local, u, v);
    float cos_el = local.y, sin_el = sqrtf(clampf(1.0f - cos_el * cos_el, 0.0f, 1.0f));
    if (sin_el <= 0.0f) return 0.0f;
    int x = (int)clampf(u * L.env_w, 0.0f, (float)(L.env_w - 1)), y = (int)clampf(v * L.env_h, 0.0f, (float)(L.env_h - 1));
    float pdf = sc.env_marg[L.env_pdf_marg - sc.env_marg_first + y] * sc.env_tables[L.env_pdf_rows + (int64_t)y * L.env_w + x] * L.env_w * L.env_h;
    return pdf / (2.0f * kPi * kPi * sin_el);
}

template <class Ft = FeatAll>
LJ_HD f3 light_emission(const DScene &sc, const DLight &L, f3 view_dir, f3 light_normal) {
    if (!Ft::envmap || L.kind == 0) {  // diffuse_area_light.inl:15-20
        if (dot(light_normal, view_dir) <= 0.0f) return mk3(0, 0, 0);
        return ld3(L.intensity);
    }
    f3 w = xform_vector9(L.to_local, -view_dir);  // envmap.inl:44-73 (the footprint it derives is <= 0 => mip level 0)
    float u, v; envmap_dir_to_uv(w, u, v);
    float dudwx = -w.z / (w.x * w.x + w.z * w.z), dudwz = w.x / (w.x * w.x + w.z * w.z);
    float dvdwy = -1.0f / sqrtf(fmaxf(1.0f - w.y * w.y, 0.0f));
    float footprint = fminf(sqrtf(dudwx * dudwx + dudwz * dudwz), dvdwy);
    return eval_texture<FeatAll>(sc, L.values, true, u, v, footprint) * L.scale;
}

// ------------------------------------------------------------------ BSDFs (materials/*.inl, microfacet.h)
LJ_HD f3 sample_cos_hemisphere(float r0, float r1) {  // material.cpp:4-11
    float sn, cs, tmp = sqrtf(clampf(1.0f - r1, 0.0f, 1.0f));
    sincos_2pi(r0, sn, cs);
    return mk3(cs * tmp, sn * tmp, sqrtf(clampf(r1, 0.0f, 1.0f)));
}
LJ_HD float fresnel_dielectric(float n_dot_i, float eta) {  // microfacet.h:34-56
    float n_dot_t_sq = 1.0f - (1.0f - n_dot_i * n_dot_i) / (eta * eta);
    if (n_dot_t_sq < 0.0f) return 1.0f;
    float ni = fabsf(n_dot_i), nt = sqrtf(n_dot_t_sq);
    float rs = (ni - eta * nt) / (ni + eta * nt), rp = (eta * ni - nt) / (eta * ni + nt);
    return (rs * rs + rp * rp) * 0.5f;
}
// GTR2 (microfacet.h:58-63) from the half vector in the shading frame.  The reference's 1 + (a2 - 1) cos^2 is
// a2 cos^2 + sin^2; written with sin^2 = hx^2 + hy^2 it keeps its precision in float when the half vector is within
// 1e-4 of the normal — at the smallest roughness the reference allows (0.01 -> alpha 1e-4) the cos^2 form rounds to
// exactly 0 there and the BSDF to inf / inf.
LJ_HD float GTR2(f3 h_local, float roughness) {
    float alpha = roughness * roughness, a2 = alpha * alpha;
    float t = a2 * h_local.z * h_local.z + (h_local.x * h_local.x + h_local.y * h_local.y);
    return a2 / (kPi * t * t);
}
LJ_HD float smith_masking_gtr2(f3 v_local, float roughness) {  // microfacet.h:75-81
    float alpha = roughness * roughness, a2 = alpha * alpha;
    f3 v2 = v_local * v_local;
    float Lambda = (-1.0f + sqrtf(1.0f + (v2.x * a2 + v2.y * a2) / v2.z)) * 0.5f;
    return 1.0f / (1.0f + Lambda);
}
LJ_HD f3 sample_visible_normals(f3 local_dir_in, float alpha, float r0, float r1) {  // microfacet.h:85-114
    bool flipped = local_dir_in.z < 0.0f;
    if (flipped) local_dir_in = -local_dir_in;
    f3 hemi = normalize(mk3(alpha * local_dir_in.x, alpha * local_dir_in.y, local_dir_in.z));
    float r = sqrtf(r0), sn, cs;
    sincos_2pi(r1, sn, cs);
    float t1 = r * cs, t2 = r * sn;
    float s = (1.0f + hemi.z) * 0.5f;
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    f3 disk = mk3(t1, t2, sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)));
    f3 hn = to_world(make_frame(hemi), disk);
    f3 n = normalize(mk3(alpha * hn.x, alpha * hn.y, fmaxf(0.0f, hn.z)));
    return flipped ? -n : n;
}


template <class Ft = FeatAll>
LJ_HD f3 tex3(const DScene &sc, const DMaterial &m, int slot, const DVertex &vx) { return eval_texture<Ft>(sc, m.tex[slot], true, vx.u, vx.v, vx.uv_screen_size); }
template <class Ft = FeatAll>
LJ_HD float tex1(const DScene &sc, const DMaterial &m, int slot, const DVertex &vx) { return eval_texture<Ft>(sc, m.tex[slot], false, vx.u, vx.v, vx.uv_screen_size).x; }

// ---- Disney family helpers (disney_metal.inl:3-50, disney_clearcoat.inl:3-16); pow(x, 5) / pow(x, 2) spelled as products
LJ_HD float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
LJ_HD float sqr(float x) { return x * x; }
LJ_HD float smithG_GGX_aniso(float NdotW, float WdotX, float WdotY, float ax, float ay) {
    float lambda = 0.5f * (sqrtf(1.0f + (sqr(WdotX * ax) + sqr(WdotY * ay)) / sqr(NdotW)) - 1.0f);
    return 1.0f / (1.0f + lambda);
}
LJ_HD float GTR2_aniso(float ax, float ay, const Frame3 &frame, f3 h) {
    float hlx2 = sqr(dot(frame.x, h)), hly2 = sqr(dot(frame.y, h)), hlz2 = sqr(dot(frame.n, h));
    return 1.0f / (kPi * ax * ay * sqr(hlx2 / (ax * ax) + hly2 / (ay * ay) + hlz2));
}
LJ_HD f3 sample_visible_normals_aniso(f3 local_dir_in, float ax, float ay, float r0, float r1) {
    bool flipped = local_dir_in.z < 0.0f;
    if (flipped) local_dir_in = -local_dir_in;
    f3 hemi = normalize(mk3(ax * local_dir_in.x, ay * local_dir_in.y, local_dir_in.z));
    float r = sqrtf(r0), sn, cs;
    sincos_2pi(r1, sn, cs);
    float t1 = r * cs, t2 = r * sn;
    float s = (1.0f + hemi.z) * 0.5f;
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    f3 disk = mk3(t1, t2, sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)));
    f3 hn = to_world(make_frame(hemi), disk);
    f3 n = normalize(mk3(ax * hn.x, ay * hn.y, fmaxf(0.0f, hn.z)));
    return flipped ? -n : n;
}
LJ_HD float clearcoat_schlick_fresnel(f3 h, f3 dir_out) {
    const float R_0 = 0.04f;  // ((1.5 - 1) / (1.5 + 1))^2
    return R_0 + (1.0f - R_0) * pow5(1.0f - fabsf(dot(h, dir_out)));
}
LJ_HD float compute_Dc(float clearcoat_gloss, float hlz2) {
    float a = (1.0f - clearcoat_gloss) * 0.1f + clearcoat_gloss * 0.001f, a2 = a * a;
    return (a2 - 1.0f) / (kPi * logf(a2) * (1.0f + (a2 - 1.0f) * hlz2));
}
LJ_HD void aniso_alphas(float roughness, float anisotropic, float &ax, float &ay) {
    float aspect = sqrtf(1.0f - 0.9f * anisotropic);
    ax = fmaxf(0.0001f, roughness * roughness / aspect); ay = fmaxf(0.0001f, roughness * roughness * aspect);
}
LJ_HD Frame3 frame_two_sided(const DVertex &vx, f3 dir_in) {  // roughdielectric.inl:6-9
    return (dot(vx.frame.n, dir_in) * dot(vx.gn, dir_in) < 0.0f) ? flip(vx.frame) : vx.frame;
}
LJ_HD f3 color_tint(f3 base_color) { float l = luminance(base_color); return l <= 0.0f ? mk3(1, 1, 1) : base_color / l; }
LJ_HD f3 disney_diffuse_lobe(f3 base_color, float roughness, float subsurface, const Frame3 &frame, f3 dir_in, f3 dir_out) {  // disney_diffuse.inl:19-39
    f3 h = normalize(dir_in + dir_out);
    float h_dot_out = dot(h, dir_out), n_dot_in = dot(frame.n, dir_in), n_dot_out = dot(frame.n, dir_out);
    float FD90 = 0.5f + 2.0f * roughness * h_dot_out * h_dot_out;
    float FD_in = 1.0f + (FD90 - 1.0f) * (1.0f - pow5(n_dot_in)), FD_out = 1.0f + (FD90 - 1.0f) * (1.0f - pow5(n_dot_out));
    f3 f_d = base_color * (FD_in * FD_out * fabsf(n_dot_out) * kInvPi);
    float FSS90 = roughness * h_dot_out * h_dot_out;
    float FSS_in = 1.0f + (FSS90 - 1.0f) * (1.0f - pow5(n_dot_in)), FSS_out = 1.0f + (FSS90 - 1.0f) * (1.0f - pow5(n_dot_out));
    f3 f_ss = base_color * (1.25f * (FSS_in * FSS_out * (1.0f / (fabsf(n_dot_in) + fabsf(n_dot_out)) - 0.5f) + 0.5f) * fabsf(n_dot_out) * kInvPi);
    return f_d * (1.0f - subsurface) + f_ss * subsurface;
}
// disney_glass.inl:3-135 (eval and pdf share everything but the last line)
LJ_HD void disney_glass_lobe(f3 base_color, float roughness_raw, float anisotropic, float bsdf_eta, const DVertex &vx, f3 dir_in, f3 dir_out, f3 &f, float &pdf) {
    bool reflect = dot(vx.gn, dir_in) * dot(vx.gn, dir_out) > 0.0f;
    Frame3 frame = frame_two_sided(vx, dir_in);
    float eta = dot(vx.gn, dir_in) > 0.0f ? bsdf_eta : 1.0f / bsdf_eta;
    f3 h = reflect ? normalize(dir_in + dir_out) : normalize(dir_in + dir_out * eta);
    if (dot(h, frame.n) < 0.0f) h = -h;
    float roughness = clampf(roughness_raw, 0.01f, 1.0f);
    float ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
    float h_dot_in = dot(h, dir_in);
    float F = fresnel_dielectric(h_dot_in, eta);
    float D = GTR2_aniso(ax, ay, frame, h);
    float G = smithG_GGX_aniso(dot(dir_in, frame.n), dot(dir_in, frame.x), dot(dir_in, frame.y), ax, ay);
    float n_dot_in = dot(frame.n, dir_in);
    if (reflect) {
        pdf = (F * D * G) / (4.0f * fabsf(n_dot_in));
        f = base_color * pdf;
        return;
    }
    float h_dot_out = dot(h, dir_out);
    float sqrt_denom = h_dot_in + eta * h_dot_out;
    float dh_dout = eta * eta * h_dot_out / (sqrt_denom * sqrt_denom);
    pdf = (1.0f - F) * D * G * fabsf(dh_dout * h_dot_in / n_dot_in);
    f3 sq = mk3(sqrtf(fmaxf(base_color.x, 0.0f)), sqrtf(fmaxf(base_color.y, 0.0f)), sqrtf(fmaxf(base_color.z, 0.0f)));
    f = sq * ((1.0f - F) * D * G * fabsf(h_dot_out * h_dot_in) / (fabsf(n_dot_in) * sqrt_denom * sqrt_denom));
}
LJ_HD f3 sample_clearcoat_half(float clearcoat_gloss, float r0, float r1) {  // disney_clearcoat.inl:85-97
    float a = (1.0f - clearcoat_gloss) * 0.1f + clearcoat_gloss * 0.001f, a2 = a * a;
    float cos_e = sqrtf((1.0f - powf(a2, 1.0f - r0)) / (1.0f - a2));
    float sin_e = sqrtf(fmaxf(0.0f, 1.0f - cos_e * cos_e));  // sin(acos(c))
    float sn, cs;
    sincos_2pi(r1, sn, cs);
    return normalize(mk3(sin_e * cs, sin_e * sn, cos_e));
}

struct BsdfSample { f3 dir_out; float eta, roughness; bool valid; };

// shared tail of the dielectric samplers (roughdielectric.inl:150-176, disney_glass.inl:180-205, disney_bsdf.inl:510-535)
LJ_HD void sample_dielectric_tail(f3 dir_in, f3 h, const Frame3 &frame, float eta, float roughness, float rnd, BsdfSample &s) {
    if (dot(h, frame.n) < 0.0f) h = -h;
    float h_dot_in = dot(h, dir_in);
    float F = fresnel_dielectric(h_dot_in, eta);
    s.roughness = roughness;
    if (rnd <= F) { s.dir_out = normalize(-dir_in + h * (2.0f * dot(dir_in, h))); s.eta = 0.0f; s.valid = true; return; }
    float h_dot_out_sq = 1.0f - (1.0f - h_dot_in * h_dot_in) / (eta * eta);
    if (h_dot_out_sq <= 0.0f) { s.valid = false; return; }
    if (h_dot_in < 0.0f) h = -h;
    float h_dot_out = sqrtf(h_dot_out_sq);
    s.dir_out = -dir_in / eta + h * (fabsf(h_dot_in) / eta - h_dot_out); s.eta = eta; s.valid = true;
}


// eval (BSDF * |cos|) and pdf together: the integrator always needs both for the same pair of directions
// (path_tracing.h:166,187 and :251-252).
template <class Ft = FeatAll>
LJ_HD void bsdf_eval_pdf(const DScene &sc, const DMaterial &m, f3 dir_in, f3 dir_out, const DVertex &vx, f3 &f, float &pdf) {
    f = mk3(0, 0, 0); pdf = 0.0f;
    const bool above = !(dot(vx.gn, dir_in) < 0.0f || dot(vx.gn, dir_out) < 0.0f);
    if (Ft::kind(2) && m.kind == 2) {  // roughdielectric.inl:3-88 (TransportDirection::TO_LIGHT, the only one path_tracing uses)
        bool reflect = dot(vx.gn, dir_in) * dot(vx.gn, dir_out) > 0.0f;
        Frame3 frame = frame_two_sided(vx, dir_in);
        float eta = dot(vx.gn, dir_in) > 0.0f ? m.eta : 1.0f / m.eta;
        f3 h = reflect ? normalize(dir_in + dir_out) : normalize(dir_in + dir_out * eta);
        if (dot(h, frame.n) < 0.0f) h = -h;
        float roughness = clampf(tex1<Ft>(sc, m, 2, vx), 0.01f, 1.0f);
        float h_dot_in = dot(h, dir_in);
        float F = fresnel_dielectric(h_dot_in, eta);
        float D = GTR2(to_local(frame, h), roughness);
        float G_in = smith_masking_gtr2(to_local(frame, dir_in), roughness);
        float G = G_in * smith_masking_gtr2(to_local(frame, dir_out), roughness);
        float n_dot_in = dot(frame.n, dir_in);
        if (reflect) {
            f = tex3<Ft>(sc, m, 0, vx) * ((F * D * G) / (4.0f * fabsf(n_dot_in)));
            pdf = (F * D * G_in) / (4.0f * fabsf(n_dot_in));
        } else {
            float h_dot_out = dot(h, dir_out);
            float sqrt_denom = h_dot_in + eta * h_dot_out;
            f = tex3<Ft>(sc, m, 1, vx) * (((1.0f / (eta * eta)) * (1.0f - F) * D * G * eta * eta * fabsf(h_dot_out * h_dot_in)) / (fabsf(n_dot_in) * sqrt_denom * sqrt_denom));
            float dh_dout = eta * eta * h_dot_out / (sqrt_denom * sqrt_denom);
            pdf = (1.0f - F) * D * G_in * fabsf(dh_dout * h_dot_in / n_dot_in);
        }
        return;
    }
    if (Ft::kind(5) && m.kind == 5) {  // disney_glass.inl
        disney_glass_lobe(tex3<Ft>(sc, m, 0, vx), tex1<Ft>(sc, m, 1, vx), tex1<Ft>(sc, m, 2, vx), m.eta, vx, dir_in, dir_out, f, pdf);
        return;
    }
    if (Ft::kind(8) && m.kind == 8) {  // disney_bsdf.inl:3-372
        f3 base_color = tex3<Ft>(sc, m, 0, vx);
        float specular_transmission = tex1<Ft>(sc, m, 1, vx), metallic = tex1<Ft>(sc, m, 2, vx), subsurface = tex1<Ft>(sc, m, 3, vx), specular = tex1<Ft>(sc, m, 4, vx);
        float roughness_raw = tex1<Ft>(sc, m, 5, vx), specular_tint = tex1<Ft>(sc, m, 6, vx), anisotropic = tex1<Ft>(sc, m, 7, vx), sheen = tex1<Ft>(sc, m, 8, vx);
        float sheen_tint = tex1<Ft>(sc, m, 9, vx), clearcoat = tex1<Ft>(sc, m, 10, vx), clearcoat_gloss = tex1<Ft>(sc, m, 11, vx);
        const bool inside = dot(vx.gn, dir_in) < 0.0f;
        const bool reflect = dot(vx.gn, dir_in) * dot(vx.gn, dir_out) > 0.0f;
        f3 f_glass; float glass_pdf;
        disney_glass_lobe(base_color, roughness_raw, anisotropic, m.eta, vx, dir_in, dir_out, f_glass, glass_pdf);
        float diffuse_weight = (1.0f - metallic) * (1.0f - specular_transmission);
        float metal_weight = 1.0f - specular_transmission * (1.0f - metallic);
        float glass_weight = (1.0f - metallic) * specular_transmission;
        float clearcoat_weight = 0.25f * clearcoat;
        f = f_glass * glass_weight;
        bool pdf_zero = false;
        if (inside) { diffuse_weight = metal_weight = clearcoat_weight = 0.0f; if (glass_weight > 0.0f) glass_weight = 1.0f; else pdf_zero = true; }
        float wsum = diffuse_weight + metal_weight + glass_weight + clearcoat_weight;
        Frame3 frame = vx.frame;
        if (dot(frame.n, dir_in) < 0.0f) frame = flip(frame);
        f3 h = normalize(dir_in + dir_out);
        float roughness = clampf(roughness_raw, 0.01f, 1.0f);
        float ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
        float Dm = GTR2_aniso(ax, ay, frame, h);
        float n_dot_in = dot(dir_in, frame.n);
        float Gin = smithG_GGX_aniso(n_dot_in, dot(dir_in, frame.x), dot(dir_in, frame.y), ax, ay);
        float n_dot_h = dot(frame.n, h);
        float Dc = compute_Dc(clearcoat_gloss, n_dot_h * n_dot_h);
        if (!pdf_zero) {
            if (reflect) {
                float diffuse_pdf = fmaxf(dot(frame.n, dir_out), 0.0f) * kInvPi;
                float metal_pdf = Dm * Gin / (4.0f * fabsf(n_dot_in));
                float clearcoat_pdf = Dc * fabsf(n_dot_h) / (4.0f * fabsf(dot(h, dir_out)));
                pdf = (diffuse_weight * diffuse_pdf + metal_weight * metal_pdf + clearcoat_weight * clearcoat_pdf + glass_weight * glass_pdf) / wsum;
            } else pdf = glass_weight * glass_pdf / wsum;
        }
        if (!inside && dot(vx.gn, dir_out) >= 0.0f) {  // the four reflective lobes (disney_bsdf.inl:24-129)
            f3 f_diffuse = disney_diffuse_lobe(base_color, roughness_raw, subsurface, frame, dir_in, dir_out);
            float h_dot_out = dot(h, dir_out);
            f3 C_tint = color_tint(base_color);
            f3 Ks = mk3(1, 1, 1) * (1.0f - specular_tint) + C_tint * specular_tint;
            f3 C0 = Ks * (specular * 0.04f * (1.0f - metallic)) + base_color * metallic;
            f3 Fm = C0 + (mk3(1, 1, 1) - C0) * pow5(1.0f - h_dot_out);  // no fabs here, as written (disney_bsdf.inl:77)
            float Gout = smithG_GGX_aniso(dot(dir_out, frame.n), dot(dir_out, frame.x), dot(dir_out, frame.y), ax, ay);
            f3 f_metal = Fm * (Dm * Gin * Gout / (4.0f * fabsf(n_dot_in)));
            float f_clearcoat = 0.0f;
            if (n_dot_h > 0.0f) {
                float G = smith_masking_gtr2(to_local(frame, dir_in), 0.5f) * smith_masking_gtr2(to_local(frame, dir_out), 0.5f);
                f_clearcoat = clearcoat_schlick_fresnel(h, dir_out) * Dc * G / (4.0f * fabsf(n_dot_in));
            }
            f3 C_sheen = mk3(1, 1, 1) * (1.0f - sheen_tint) + C_tint * sheen_tint;
            f3 f_sheen = C_sheen * (pow5(1.0f - fabsf(h_dot_out)) * fabsf(dot(frame.n, dir_out)));
            f = f + f_diffuse * ((1.0f - specular_transmission) * (1.0f - metallic)) + f_sheen * ((1.0f - metallic) * sheen)
                  + f_metal * (1.0f - specular_transmission * (1.0f - metallic)) + mk3(1, 1, 1) * (0.25f * clearcoat * f_clearcoat);
        }
        return;
    }
    if (!above) return;  // one-sided materials: no light below the surface (lambertian.inl:2-6 and the like)
    Frame3 frame = vx.frame;
    if (dot(frame.n, dir_in) < 0.0f) frame = flip(frame);
    float n_dot_out = dot(frame.n, dir_out);
    if ((Ft::kind(3) && m.kind == 3) || (Ft::kind(7) && m.kind == 7)) {  // disney_diffuse.inl:1-58, disney_sheen.inl:3-46
        pdf = fmaxf(n_dot_out, 0.0f) * kInvPi;
        if (m.kind == 3) f = disney_diffuse_lobe(tex3<Ft>(sc, m, 0, vx), tex1<Ft>(sc, m, 1, vx), tex1<Ft>(sc, m, 2, vx), frame, dir_in, dir_out);
        else {
            f3 h = normalize(dir_in + dir_out);
            float sheen_tint = tex1<Ft>(sc, m, 1, vx);
            f3 C_sheen = mk3(1, 1, 1) * (1.0f - sheen_tint) + color_tint(tex3<Ft>(sc, m, 0, vx)) * sheen_tint;
            f = C_sheen * (pow5(1.0f - fabsf(dot(h, dir_out))) * fabsf(n_dot_out));
        }
        return;
    }
    if (Ft::kind(4) && m.kind == 4) {  // disney_metal.inl:52-126
        f3 base_color = tex3<Ft>(sc, m, 0, vx);
        float roughness = clampf(tex1<Ft>(sc, m, 1, vx), 0.01f, 1.0f), anisotropic = tex1<Ft>(sc, m, 2, vx);
        f3 h = normalize(dir_in + dir_out);
        f3 Fm = base_color + (mk3(1, 1, 1) - base_color) * pow5(1.0f - fabsf(dot(h, dir_out)));
        float ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
        float Dm = GTR2_aniso(ax, ay, frame, h);
        float n_dot_in = dot(dir_in, frame.n);
        float Gin = smithG_GGX_aniso(n_dot_in, dot(dir_in, frame.x), dot(dir_in, frame.y), ax, ay);
        float Gout = smithG_GGX_aniso(n_dot_out, dot(dir_out, frame.x), dot(dir_out, frame.y), ax, ay);
        pdf = Dm * Gin / (4.0f * fabsf(n_dot_in));
        f = Fm * (pdf * Gout);
        return;
    }
    if (Ft::kind(6) && m.kind == 6) {  // disney_clearcoat.inl:18-66
        f3 h = normalize(dir_in + dir_out);
        float n_dot_h = dot(frame.n, h);
        float D = compute_Dc(tex1<Ft>(sc, m, 0, vx), n_dot_h * n_dot_h);
        pdf = D * fabsf(n_dot_h) / (4.0f * fabsf(dot(h, dir_out)));
        if (n_dot_h > 0.0f) {
            float G = smith_masking_gtr2(to_local(frame, dir_in), 0.5f) * smith_masking_gtr2(to_local(frame, dir_out), 0.5f);
            float v = clearcoat_schlick_fresnel(h, dir_out) * D * G / (4.0f * fabsf(dot(frame.n, dir_in)));
            f = mk3(v, v, v);
        }
        return;
    }
    if (Ft::kind(0) && m.kind == 0) {  // lambertian.inl:1-33
        float c = fmaxf(n_dot_out, 0.0f);
        f = tex3<Ft>(sc, m, 0, vx) * (c * kInvPi);
        pdf = c * kInvPi;
        return;
    }
    if (Ft::kind(1) && m.kind == 1) {  // roughplastic.inl:3-108
        f3 h = normalize(dir_in + dir_out);
        float n_dot_h = dot(frame.n, h), n_dot_in = dot(frame.n, dir_in);
        if (n_dot_out <= 0.0f || n_dot_h <= 0.0f) return;
        f3 Kd = tex3<Ft>(sc, m, 0, vx), Ks = tex3<Ft>(sc, m, 1, vx);
        float roughness = clampf(tex1<Ft>(sc, m, 2, vx), 0.01f, 1.0f);
        float F_o = fresnel_dielectric(dot(h, dir_out), m.eta);
        float D = GTR2(to_local(frame, h), roughness);
        float G_in = smith_masking_gtr2(to_local(frame, dir_in), roughness);
        float G = G_in * smith_masking_gtr2(to_local(frame, dir_out), roughness);
        f3 spec = Ks * ((G * F_o * D) / (4.0f * n_dot_in * n_dot_out));
        float F_i = fresnel_dielectric(dot(h, dir_in), m.eta);
        f3 diff = Kd * ((1.0f - F_o) * (1.0f - F_i) * kInvPi);
        f = (spec + diff) * n_dot_out;
        float lS = luminance(Ks), lR = luminance(Kd);
        if (lS + lR <= 0.0f) return;  // pdf stays 0 (roughplastic.inl:88-90)
        float spec_prob = lS / (lS + lR), diff_prob = 1.0f - spec_prob;
        pdf = spec_prob * (G_in * D) / (4.0f * n_dot_in) + diff_prob * n_dot_out * kInvPi;
        return;
    }
}

template <class Ft = FeatAll>
LJ_HD BsdfSample bsdf_sample(const DScene &sc, const DMaterial &m, f3 dir_in, const DVertex &vx, float r0, float r1, float rw) {
    BsdfSample s; s.valid = false; s.eta = 0.0f; s.roughness = 1.0f; s.dir_out = mk3(0, 0, 0);
    if (Ft::kind(2) && m.kind == 2) {  // roughdielectric.inl:90-177
        float eta = dot(vx.gn, dir_in) > 0.0f ? m.eta : 1.0f / m.eta;
        Frame3 fr = frame_two_sided(vx, dir_in);
        float roughness = clampf(tex1<Ft>(sc, m, 2, vx), 0.01f, 1.0f);
        f3 h = to_world(fr, sample_visible_normals(to_local(fr, dir_in), roughness * roughness, r0, r1));
        sample_dielectric_tail(dir_in, h, fr, eta, roughness, rw, s);
        return s;
    }
    if (Ft::kind(5) && m.kind == 5) {  // disney_glass.inl:137-205
        Frame3 fr = frame_two_sided(vx, dir_in);
        float eta = dot(vx.gn, dir_in) > 0.0f ? m.eta : 1.0f / m.eta;
        float roughness = clampf(tex1<Ft>(sc, m, 1, vx), 0.01f, 1.0f);
        float ax, ay; aniso_alphas(roughness, tex1<Ft>(sc, m, 2, vx), ax, ay);
        f3 h = to_world(fr, sample_visible_normals_aniso(to_local(fr, dir_in), ax, ay, r0, r1));
        sample_dielectric_tail(dir_in, h, fr, eta, roughness, rw, s);
        return s;
    }
    if (Ft::kind(8) && m.kind == 8) {  // disney_bsdf.inl:374-572
        float specular_transmission = tex1<Ft>(sc, m, 1, vx), metallic = tex1<Ft>(sc, m, 2, vx), anisotropic = tex1<Ft>(sc, m, 7, vx);
        float clearcoat = tex1<Ft>(sc, m, 10, vx);
        float eta = dot(vx.gn, dir_in) > 0.0f ? m.eta : 1.0f / m.eta;
        float diffuse_weight = (1.0f - metallic) * (1.0f - specular_transmission);
        float metal_weight = 1.0f - specular_transmission * (1.0f - metallic);
        float glass_weight = (1.0f - metallic) * specular_transmission;
        float clearcoat_weight = 0.25f * clearcoat;
        if (dot(vx.gn, dir_in) < 0.0f) {
            diffuse_weight = metal_weight = clearcoat_weight = 0.0f;
            if (glass_weight > 0.0f) glass_weight = 1.0f;
            else { s.valid = true; return s; }  // zero-direction record, not nullopt (disney_bsdf.inl:418-420); its pdf is 0
        }
        float wsum = diffuse_weight + metal_weight + glass_weight + clearcoat_weight;
        diffuse_weight /= wsum; metal_weight /= wsum; glass_weight /= wsum;
        Frame3 fr = vx.frame;
        if (dot(fr.n, dir_in) < 0.0f) fr = flip(fr);
        if (rw < diffuse_weight) { s.dir_out = to_world(fr, sample_cos_hemisphere(r0, r1)); s.valid = true; }
        else if (rw < diffuse_weight + metal_weight) {
            float roughness = clampf(tex1<Ft>(sc, m, 5, vx), 0.01f, 1.0f);
            float ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
            f3 h = to_world(fr, sample_visible_normals_aniso(to_local(fr, dir_in), ax, ay, r0, r1));
            s.dir_out = normalize(-dir_in + h * (2.0f * dot(dir_in, h))); s.roughness = roughness; s.valid = true;
        } else if (rw < diffuse_weight + metal_weight + glass_weight) {
            Frame3 f2 = frame_two_sided(vx, dir_in);
            float roughness = clampf(tex1<Ft>(sc, m, 5, vx), 0.01f, 1.0f);
            float ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
            f3 h = to_world(f2, sample_visible_normals_aniso(to_local(f2, dir_in), ax, ay, r0, r1));
            float rand_new = (rw - (diffuse_weight + metal_weight)) / glass_weight;
            sample_dielectric_tail(dir_in, h, f2, eta, roughness, rand_new, s);
        } else {
            f3 h = to_world(fr, sample_clearcoat_half(tex1<Ft>(sc, m, 11, vx), r0, r1));
            s.dir_out = normalize(-dir_in + h * (2.0f * dot(dir_in, h))); s.valid = true;
        }
        return s;
    }
    if (dot(vx.gn, dir_in) < 0.0f) return s;  // one-sided materials (lambertian.inl:37-40 and the like)
    Frame3 frame = vx.frame;
    if (dot(frame.n, dir_in) < 0.0f) frame = flip(frame);
    if ((Ft::kind(3) && m.kind == 3) || (Ft::kind(7) && m.kind == 7)) {  // disney_diffuse.inl:60-76, disney_sheen.inl:48-62
        s.dir_out = to_world(frame, sample_cos_hemisphere(r0, r1)); s.valid = true;
        return s;
    }
    if (Ft::kind(4) && m.kind == 4) {  // disney_metal.inl:128-162
        float roughness = clampf(tex1<Ft>(sc, m, 1, vx), 0.01f, 1.0f);
        float ax, ay; aniso_alphas(roughness, tex1<Ft>(sc, m, 2, vx), ax, ay);
        f3 h = to_world(frame, sample_visible_normals_aniso(to_local(frame, dir_in), ax, ay, r0, r1));
        s.dir_out = normalize(-dir_in + h * (2.0f * dot(dir_in, h))); s.roughness = roughness; s.valid = true;
        return s;
    }
    if (Ft::kind(6) && m.kind == 6) {  // disney_clearcoat.inl:68-106
        f3 h = to_world(frame, sample_clearcoat_half(tex1<Ft>(sc, m, 0, vx), r0, r1));
        s.dir_out = normalize(-dir_in + h * (2.0f * dot(dir_in, h))); s.valid = true;
        return s;
    }
    if (Ft::kind(0) && m.kind == 0) {  // lambertian.inl:35-50
        s.dir_out = to_world(frame, sample_cos_hemisphere(r0, r1)); s.valid = true;
        return s;
    }
    if (Ft::kind(1) && m.kind == 1) {  // roughplastic.inl:110-161
        f3 Ks = tex3<Ft>(sc, m, 1, vx), Kd = tex3<Ft>(sc, m, 0, vx);
        float lS = luminance(Ks), lR = luminance(Kd);
        if (lS + lR <= 0.0f) return s;
        float spec_prob = lS / (lS + lR);
        if (rw < spec_prob) {
            float roughness = clampf(tex1<Ft>(sc, m, 2, vx), 0.01f, 1.0f);
            f3 hm = to_world(frame, sample_visible_normals(to_local(frame, dir_in), roughness * roughness, r0, r1));
            s.dir_out = normalize(-dir_in + hm * (2.0f * dot(dir_in, hm)));
            s.roughness = roughness;
        } else s.dir_out = to_world(frame, sample_cos_hemisphere(r0, r1));
        s.valid = true;
        return s;
    }
    return s;
}

LJ_HD bool material_supported(int kind) { return kind >= 0 && kind <= 8; }

// ------------------------------------------------------------------ the per-path record the kernels move
struct PathState {
    f3 org, dir;            // extension ray that was just traced
    float ht, hu, hv; int32_t hcode;   // its hit (hcode: (gprim+1) | HIT_VIS_BIT)
    f3 sdir; float stfar;   // pending NEE shadow ray (written for the next extend)
    f3 W; float rr, p2;
    f3 rad, nee;
    uint32_t sample; uint64_t rng;
    float eta_scale, spread;
    uint32_t flags;
};

struct ShadeCounters { uint32_t bounces, closest, shadow, done; };

// Camera-sample generation: path_tracing.h:10-14 + render.cpp:82 (per-sample stream variant).
LJ_HD void generate_path(const DScene &sc, const DPass &pass, uint32_t sample_id, PathState &ps) {
    const uint32_t p = fast_div(sample_id, pass.by_spp), s = sample_id - p * pass.spp;
    const uint32_t pixel = pass.pixel_list[p];
    const int y = (int)fast_div(pixel, pass.by_width), x = (int)(pixel - (uint32_t)y * (uint32_t)sc.cam.width);
    const uint64_t stream = (uint64_t)pixel * pass.spp + s;
    const uint64_t inc = pcg32_inc(stream);
    uint64_t st = pcg32_init(stream, pass.seed);
    // the reference's g++ build gives the first draw to the y jitter and the second to x
    const float jy = pcg32_real(st, inc);
    const float jx = pcg32_real(st, inc);
    ps.org = ld3(sc.cam.org);
    ps.dir = camera_primary_dir(sc.cam, x, y, jx, jy);
    ps.sdir = mk3(0, 0, 0); ps.stfar = 0.0f;
    ps.W = mk3(1, 1, 1); ps.rr = 1.0f; ps.p2 = -1.0f;
    ps.rad = mk3(0, 0, 0); ps.nee = mk3(0, 0, 0);
    ps.sample = sample_id; ps.rng = st;
    ps.eta_scale = 1.0f; ps.spread = sc.init_spread;
    ps.flags = 2u;  // num_vertices so far (camera + the vertex this ray will find); the first loop iteration is 3
}

// One wavefront step for one path: everything path_tracing() does between the return of one intersect() and the
// next (path_tracing.h:58-61 for camera rays; :239-322 tail of iteration k, then :94-237 head of iteration k+1).
// Returns true if the path continues (ps holds the next extension + shadow rays), false if it is finished
// (ps.rad is the value of path_tracing() for this sample).
template <class Ft = FeatAll>
LJ_HD bool shade_path(const DScene &sc, const DPass &pass, PathState &ps, ShadeCounters &cnt) {
    const uint32_t pix_i = fast_div(ps.sample, pass.by_spp);
    const uint64_t inc = pcg32_inc((uint64_t)pass.pixel_list[pix_i] * pass.spp + (ps.sample - pix_i * pass.spp));
    // pending next-event estimation of the previous vertex (path_tracing.h:207)
    if (ps.hcode & HIT_VIS_BIT) ps.rad = ps.rad + ps.nee;
    const bool primary = ps.p2 < 0.0f;
    if (ps.flags & PF_NO_EXT) return false;  // sample_bsdf failed or p2 <= 0 (path_tracing.h:220-223,253-256)
    const int gprim = (ps.hcode & 0x3fffffff) - 1;
    const uint32_t nv_prev = ps.flags & 0xffffu;
    if (gprim < 0) {  // miss: environment map (path_tracing.h:17-27 camera rays, :284-302 bounces)
        if (Ft::envmap && sc.envmap_light_id >= 0) {
            const DLight &E = sc.lights[sc.envmap_light_id];
            f3 L = light_emission<Ft>(sc, E, -ps.dir, mk3(0, 0, 0));
            if (primary) ps.rad = ps.rad + L;
            else {
                float p1 = E.pmf * pdf_point_on_light<Ft>(sc, E, mk3(0, 0, 0), -ps.dir, ps.org);
                float p2 = ps.p2;  // G = 1
                float w2 = (p2 * p2) / (p1 * p1 + p2 * p2);
                ps.rad = ps.rad + ps.W * L * w2;
            }
        }
        return false;
    }
    DVertex vx = build_vertex(sc, ps.org, ps.dir, ps.ht, ps.hu, ps.hv, gprim, primary ? ps.spread : 0.0f);
    if (vx.light_id >= 0) {  // hit an emitter (path_tracing.h:58-61, :268-283)
        const DLight &EL = sc.lights[vx.light_id];
        f3 L = light_emission<Ft>(sc, EL, -ps.dir, vx.gn);
        if (primary) ps.rad = ps.rad + L;
        else {
            f3 dv = vx.position - ps.org;
            float G = fabsf(dot(ps.dir, vx.gn)) / dot(dv, dv);
            float p2 = ps.p2 * G;
            float p1 = EL.pmf * pdf_point_on_light<Ft>(sc, EL, vx.position, vx.gn, ps.org);
            float w2 = (p2 * p2) / (p1 * p1 + p2 * p2);
            ps.rad = ps.rad + ps.W * L * w2;
        }
    }
    if (ps.flags & PF_DYING) return false;  // Russian roulette said stop (path_tracing.h:314-317)
    // loop header of the next iteration (path_tracing.h:66)
    const uint32_t nv = nv_prev + 1;
    if (!(sc.max_depth == -1 || (int)nv <= sc.max_depth + 1)) return false;
    cnt.bounces++;
    f3 thr = primary ? mk3(1, 1, 1) : ps.W / ps.rr;  // path_tracing.h:322
    const DMaterial &mat = sc.materials[vx.material_id];
    const f3 dir_view = -ps.dir;

    // ---- next event estimation (path_tracing.h:98-207)
    float lu0 = pcg32_real(ps.rng, inc), lu1 = pcg32_real(ps.rng, inc);
    float light_w = pcg32_real(ps.rng, inc), shape_w = pcg32_real(ps.rng, inc);
    int light_id = sample_cdf(sc.light_cdf, sc.n_lights, light_w);
    const DLight &Lt = sc.lights[light_id];
    LightSample pl = sample_point_on_light<Ft>(sc, Lt, vx.position, lu0, lu1, shape_w);
    ps.nee = mk3(0, 0, 0); ps.sdir = mk3(0, 0, 0); ps.stfar = 0.0f;
    {
        float G; f3 dir_light; float tfar;
        if (!Ft::envmap || Lt.kind == 0) {
            if (Ft::sphere_lights && Lt.is_sphere) {   // direction and distance from the double sample point (see LightSample)
                const double dx = pl.dpos[0] - (double)vx.position.x, dy = pl.dpos[1] - (double)vx.position.y, dz = pl.dpos[2] - (double)vx.position.z;
                const double d2d = dx * dx + dy * dy + dz * dz, dd = sqrt(d2d);
                dir_light = mk3((float)(dx / dd), (float)(dy / dd), (float)(dz / dd));
                tfar = (float)((1.0 - (double)sc.eps) * dd);
                G = fmaxf(-dot(dir_light, pl.normal), 0.0f) / (float)d2d;
            } else {
                f3 dl = pl.position - vx.position;
                float d2 = dot(dl, dl), d = sqrtf(d2);
                dir_light = normalize(dl);
                tfar = (1.0f - sc.eps) * d;
                G = fmaxf(-dot(dir_light, pl.normal), 0.0f) / d2;
            }
        } else { dir_light = -pl.normal; tfar = INFINITY; G = 1.0f; }
        float p1 = Lt.pmf * pdf_point_on_light<Ft>(sc, Lt, pl.position, pl.normal, vx.position);
        if (G > 0.0f && p1 > 0.0f) {
            f3 f; float p2;
            bsdf_eval_pdf<Ft>(sc, mat, dir_view, dir_light, vx, f, p2);
            f3 Le = light_emission<Ft>(sc, Lt, -dir_light, pl.normal);
            f3 C1 = f * Le * (G / p1);
            p2 *= G;
            float w1 = (p1 * p1) / (p1 * p1 + p2 * p2);
            f3 contrib = thr * C1 * w1;
            if (contrib.x != 0.0f || contrib.y != 0.0f || contrib.z != 0.0f) {
                // the shadow ray is only worth tracing when it can add something; occlusion only ever zeroes G
                ps.nee = contrib; ps.sdir = dir_light; ps.stfar = tfar; cnt.shadow++;
            }
        }
    }
    // ---- BSDF sampling (path_tracing.h:210-237)
    float b0 = pcg32_real(ps.rng, inc), b1 = pcg32_real(ps.rng, inc), bw = pcg32_real(ps.rng, inc);
    BsdfSample bs = bsdf_sample<Ft>(sc, mat, dir_view, vx, b0, b1, bw);
    ps.org = vx.position;
    ps.flags = nv;
    if (!bs.valid) { ps.flags |= PF_NO_EXT; ps.dir = mk3(0, 0, 1); return ps.stfar > 0.0f; }
    if (bs.eta == 0.0f) ps.spread = fmaxf(ps.spread * (1.0f - bs.roughness) + 0.2f * bs.roughness, 0.0f);  // ray.h:45-51, radius == 0
    else { ps.spread = fmaxf((ps.spread / bs.eta) * (1.0f - bs.roughness) + 0.2f * bs.roughness, 0.0f); ps.eta_scale /= (bs.eta * bs.eta); }
    f3 f; float p2;
    bsdf_eval_pdf<Ft>(sc, mat, dir_view, bs.dir_out, vx, f, p2);
    if (!(p2 > 0.0f)) { ps.flags |= PF_NO_EXT; ps.dir = mk3(0, 0, 1); return ps.stfar > 0.0f; }  // path_tracing.h:253-256
    ps.dir = bs.dir_out;
    ps.W = thr * f / p2; ps.p2 = p2;
    cnt.closest++;
    // ---- Russian roulette of this iteration (path_tracing.h:311-318): decided now, applied after the hit accounting
    ps.rr = 1.0f;
    if ((int)nv - 1 >= sc.rr_depth) {
        ps.rr = fminf(max3(thr * (1.0f / ps.eta_scale)), 0.95f);
        if (pcg32_real(ps.rng, inc) > ps.rr) ps.flags |= PF_DYING;
    }
    return true;
}

} // namespace ljd
