// Per-lane shading logic of the wavefront integrator, in float: camera rays, shading info, textures, lights,
// BSDFs and the bounce step of path_tracing() re-cut into "everything between two ray casts".
// Each function cites the reference code it computes the same thing as.  Compiled for gfx950 by hipcc and, for
// CPU-side debugging of the very same source, by g++ in tests/twin (never shipped, never a fallback).
#pragma once
#include "dmath.h"

namespace ljd {

// ------------------------------------------------------------------ textures (texture.h:123-154, mipmap.h:52-89)
LJ_HD f3 texel(const DScene &sc, const DImage &img, int level, int x, int y) {
    const DMipLevel lv = img.lv[level];
    const float *p = sc.texels + lv.offset + ((int64_t)y * lv.w + x) * img.channels;
    return img.channels >= 3 ? mk3(p[0], p[1], p[2]) : mk3(p[0], p[0], p[0]);
}
LJ_HD f3 mip_lookup_level(const DScene &sc, const DImage &img, float u, float v, int level) {
    const int W = img.lv[level].w, H = img.lv[level].h;
    u = u * W - 0.5f; v = v * H - 0.5f;
    int ufi = moduloi((int)u, W), vfi = moduloi((int)v, H);
    int uci = moduloi(ufi + 1, W), vci = moduloi(vfi + 1, H);
    float uo = u - ufi, vo = v - vfi;
    f3 ff = texel(sc, img, level, ufi, vfi), fc = texel(sc, img, level, ufi, vci);
    f3 cf = texel(sc, img, level, uci, vfi), cc = texel(sc, img, level, uci, vci);
    return ff * ((1 - uo) * (1 - vo)) + fc * ((1 - uo) * vo) + cf * (uo * (1 - vo)) + cc * (uo * vo);
}
LJ_HD f3 mip_lookup(const DScene &sc, const DImage &img, float u, float v, float level) {
    if (level <= 0.0f) return mip_lookup_level(sc, img, u, v, 0);
    if (level < (float)(img.levels - 1)) {
        int fl = (int)floorf(level); fl = fl < 0 ? 0 : (fl > img.levels - 1 ? img.levels - 1 : fl);
        int cl = fl + 1 > img.levels - 1 ? img.levels - 1 : fl + 1;
        float lo = level - fl;
        return mip_lookup_level(sc, img, u, v, fl) * (1 - lo) + mip_lookup_level(sc, img, u, v, cl) * lo;
    }
    return mip_lookup_level(sc, img, u, v, img.levels - 1);
}
LJ_HD f3 eval_texture(const DScene &sc, const DTexture &t, bool spectrum, float u, float v, float footprint) {
    if (t.kind == 0) return ld3(t.value);
    float lu = modulof(u * t.uscale + t.uoffset, 1.0f), lv = modulof(v * t.vscale + t.voffset, 1.0f);
    if (t.kind == 1) {
        const DImage &img = spectrum ? sc.images3[t.texture_id] : sc.images1[t.texture_id];
        float scaled = (float)(img.lv[0].w > img.lv[0].h ? img.lv[0].w : img.lv[0].h) * fmaxf(t.uscale, t.vscale) * footprint;
        float level = log2f(fmaxf(scaled, 1e-8f));
        return mip_lookup(sc, img, lu, lv, level);
    }
    int x = 2 * moduloi((int)(lu * 2), 2) - 1, y = 2 * moduloi((int)(lv * 2), 2) - 1;
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
        ox = r * cosf(kTwoPi * r1); oy = r * sinf(kTwoPi * r1);
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

// ------------------------------------------------------------------ path vertex (intersection.cpp:38-62)
struct DVertex {
    f3 position, gn;
    Frame3 frame;
    float u, v;              // texture uv
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
        vx.u = b0 * ps.uv0[0] + bu * ps.uv1[0] + bv * ps.uv2[0];
        vx.v = b0 * ps.uv0[1] + bu * ps.uv1[1] + bv * ps.uv2[1];
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

// ------------------------------------------------------------------ lights (lights/*.inl, shapes/*.inl sampling)
struct LightSample { f3 position, normal; };

LJ_HD float sphere_one_minus_cos_max(float r, float dist_sq) {
    // 1 - sqrt(1 - r^2/d^2) without cancellation: s / (1 + sqrt(1 - s))
    float s = r * r / dist_sq;
    return s / (1.0f + sqrtf(fmaxf(0.0f, 1.0f - s)));
}

LJ_HD LightSample sample_point_on_light(const DScene &sc, const DLight &L, f3 ref, float u0, float u1, float w) {
    LightSample ls;
    if (L.kind == 0) {
        if (!L.is_sphere) {  // triangle_mesh.inl:24-38
            int tri = sample_cdf(sc.light_tri_cdf + L.cdf_first, L.tri_count, w);
            const DLightTri &T = sc.light_tris[L.tri_first + tri];
            float a = sqrtf(clampf(u0, 0.0f, 1.0f));
            float b1 = 1.0f - a, b2 = a * u1;
            ls.position = ld3(T.v0) + ld3(T.e1) * b1 + ld3(T.e2) * b2;
            ls.normal = ld3(T.n);
        } else {  // sphere.inl:156-204
            f3 center = ld3(L.center); float r = L.radius;
            f3 dc_vec = center - ref;
            float dist_sq = dot(dc_vec, dc_vec);
            if (dist_sq < r * r) {
                float z = 1.0f - 2.0f * u0;
                float r_ = sqrtf(fmaxf(0.0f, 1.0f - z * z));
                float phi = kTwoPi * u1;
                f3 offset = mk3(r_ * cosf(phi), r_ * sinf(phi), z);
                ls.position = center + offset * r; ls.normal = offset;
            } else {
                Frame3 frame = make_frame(normalize(dc_vec));
                float omc_max = sphere_one_minus_cos_max(r, dist_sq);
                float omc = u0 * omc_max;                       // 1 - cos_elevation
                float cos_elevation = 1.0f - omc;
                float sin_sq = omc * (2.0f - omc);               // 1 - cos^2, stable
                float azimuth = u1 * kTwoPi;
                float dc = sqrtf(dist_sq);
                float ds = dc * cos_elevation - sqrtf(fmaxf(0.0f, r * r - dist_sq * sin_sq));
                float cos_alpha = (dist_sq + r * r - ds * ds) / (2.0f * dc * r);
                float sin_alpha = sqrtf(fmaxf(0.0f, 1.0f - cos_alpha * cos_alpha));
                f3 n = -to_world(frame, mk3(sin_alpha * cosf(azimuth), sin_alpha * sinf(azimuth), cos_alpha));
                ls.position = n * r + center; ls.normal = n;
            }
        }
    } else {  // envmap.inl:7-20 with table_dist.cpp:116-139
        const float *cm = sc.env_tables + L.env_cdf_marg;
        int yo = sample_cdf(cm, L.env_h, u1);
        float dy = u1 - cm[yo];
        if (cm[yo + 1] - cm[yo] > 0.0f) dy /= (cm[yo + 1] - cm[yo]);
        const float *cdf = sc.env_tables + L.env_cdf_rows + (int64_t)yo * (L.env_w + 1);
        int xo = sample_cdf(cdf, L.env_w, u0);
        float dx = u0 - cdf[xo];
        if (cdf[xo + 1] - cdf[xo] > 0.0f) dx /= (cdf[xo + 1] - cdf[xo]);
        float az = ((xo + dx) / L.env_w) * kTwoPi, el = ((yo + dy) / L.env_h) * kPi;
        f3 local = mk3(sinf(az) * sinf(el), cosf(el), -cosf(az) * sinf(el));
        ls.position = mk3(0, 0, 0); ls.normal = -xform_vector9(L.to_world, local);
    }
    return ls;
}

LJ_HD void envmap_dir_to_uv(f3 local, float &u, float &v) {
    u = atan2f(local.x, -local.z) * kInvTwoPi; v = acosf(clampf(local.y, -1.0f, 1.0f)) * kInvPi;
    if (u < 0.0f) u += 1.0f;
}

LJ_HD float pdf_point_on_light(const DScene &sc, const DLight &L, f3 pos, f3 nrm, f3 ref) {
    if (L.kind == 0) {
        if (!L.is_sphere) return 1.0f / L.total_area;  // triangle_mesh.inl:44-46
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
    float pdf = sc.env_tables[L.env_pdf_marg + y] * sc.env_tables[L.env_pdf_rows + (int64_t)y * L.env_w + x] * L.env_w * L.env_h;
    return pdf / (2.0f * kPi * kPi * sin_el);
}

LJ_HD f3 light_emission(const DScene &sc, const DLight &L, f3 view_dir, f3 light_normal) {
    if (L.kind == 0) {  // diffuse_area_light.inl:15-20
        if (dot(light_normal, view_dir) <= 0.0f) return mk3(0, 0, 0);
        return ld3(L.intensity);
    }
    f3 w = xform_vector9(L.to_local, -view_dir);  // envmap.inl:44-73 (the footprint it derives is <= 0 => mip level 0)
    float u, v; envmap_dir_to_uv(w, u, v);
    float dudwx = -w.z / (w.x * w.x + w.z * w.z), dudwz = w.x / (w.x * w.x + w.z * w.z);
    float dvdwy = -1.0f / sqrtf(fmaxf(1.0f - w.y * w.y, 0.0f));
    float footprint = fminf(sqrtf(dudwx * dudwx + dudwz * dudwz), dvdwy);
    return eval_texture(sc, L.values, true, u, v, footprint) * L.scale;
}

// ------------------------------------------------------------------ BSDFs (materials/*.inl, microfacet.h)
LJ_HD f3 sample_cos_hemisphere(float r0, float r1) {  // material.cpp:4-11
    float phi = kTwoPi * r0, tmp = sqrtf(clampf(1.0f - r1, 0.0f, 1.0f));
    return mk3(cosf(phi) * tmp, sinf(phi) * tmp, sqrtf(clampf(r1, 0.0f, 1.0f)));
}
LJ_HD float fresnel_dielectric(float n_dot_i, float eta) {  // microfacet.h:34-56
    float n_dot_t_sq = 1.0f - (1.0f - n_dot_i * n_dot_i) / (eta * eta);
    if (n_dot_t_sq < 0.0f) return 1.0f;
    float ni = fabsf(n_dot_i), nt = sqrtf(n_dot_t_sq);
    float rs = (ni - eta * nt) / (ni + eta * nt), rp = (eta * ni - nt) / (eta * ni + nt);
    return (rs * rs + rp * rp) * 0.5f;
}
LJ_HD float GTR2(float n_dot_h, float roughness) {  // microfacet.h:58-63
    float alpha = roughness * roughness, a2 = alpha * alpha;
    float t = 1.0f + (a2 - 1.0f) * n_dot_h * n_dot_h;
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
    float r = sqrtf(r0), phi = kTwoPi * r1;
    float t1 = r * cosf(phi), t2 = r * sinf(phi);
    float s = (1.0f + hemi.z) * 0.5f;
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    f3 disk = mk3(t1, t2, sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)));
    f3 hn = to_world(make_frame(hemi), disk);
    f3 n = normalize(mk3(alpha * hn.x, alpha * hn.y, fmaxf(0.0f, hn.z)));
    return flipped ? -n : n;
}

struct BsdfSample { f3 dir_out; float eta, roughness; bool valid; };

LJ_HD f3 tex3(const DScene &sc, const DMaterial &m, int slot, const DVertex &vx) { return eval_texture(sc, m.tex[slot], true, vx.u, vx.v, vx.uv_screen_size); }
LJ_HD float tex1(const DScene &sc, const DMaterial &m, int slot, const DVertex &vx) { return eval_texture(sc, m.tex[slot], false, vx.u, vx.v, vx.uv_screen_size).x; }

// eval (BSDF * |cos|) and pdf together: the integrator always needs both for the same pair of directions
// (path_tracing.h:166,187 and :251-252).
LJ_HD void bsdf_eval_pdf(const DScene &sc, const DMaterial &m, f3 dir_in, f3 dir_out, const DVertex &vx, f3 &f, float &pdf) {
    f = mk3(0, 0, 0); pdf = 0.0f;
    if (dot(vx.gn, dir_in) < 0.0f || dot(vx.gn, dir_out) < 0.0f) return;  // lambertian.inl:2-6, roughplastic.inl:4-8
    Frame3 frame = vx.frame;
    if (dot(frame.n, dir_in) < 0.0f) frame = flip(frame);
    float n_dot_out = dot(frame.n, dir_out);
    if (m.kind == 0) {  // lambertian.inl:1-33
        float c = fmaxf(n_dot_out, 0.0f);
        f = tex3(sc, m, 0, vx) * (c * kInvPi);
        pdf = c * kInvPi;
        return;
    }
    if (m.kind == 1) {  // roughplastic.inl:3-108
        f3 h = normalize(dir_in + dir_out);
        float n_dot_h = dot(frame.n, h), n_dot_in = dot(frame.n, dir_in);
        if (n_dot_out <= 0.0f || n_dot_h <= 0.0f) return;
        f3 Kd = tex3(sc, m, 0, vx), Ks = tex3(sc, m, 1, vx);
        float roughness = clampf(tex1(sc, m, 2, vx), 0.01f, 1.0f);
        float F_o = fresnel_dielectric(dot(h, dir_out), m.eta);
        float D = GTR2(n_dot_h, roughness);
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

LJ_HD BsdfSample bsdf_sample(const DScene &sc, const DMaterial &m, f3 dir_in, const DVertex &vx, float r0, float r1, float rw) {
    BsdfSample s; s.valid = false; s.eta = 0.0f; s.roughness = 1.0f; s.dir_out = mk3(0, 0, 0);
    if (dot(vx.gn, dir_in) < 0.0f) return s;  // lambertian.inl:37-40, roughplastic.inl:112-115
    Frame3 frame = vx.frame;
    if (dot(frame.n, dir_in) < 0.0f) frame = flip(frame);
    if (m.kind == 0) {  // lambertian.inl:35-50
        s.dir_out = to_world(frame, sample_cos_hemisphere(r0, r1)); s.valid = true;
        return s;
    }
    if (m.kind == 1) {  // roughplastic.inl:110-161
        f3 Ks = tex3(sc, m, 1, vx), Kd = tex3(sc, m, 0, vx);
        float lS = luminance(Ks), lR = luminance(Kd);
        if (lS + lR <= 0.0f) return s;
        float spec_prob = lS / (lS + lR);
        if (rw < spec_prob) {
            float roughness = clampf(tex1(sc, m, 2, vx), 0.01f, 1.0f);
            f3 hm = to_world(frame, sample_visible_normals(to_local(frame, dir_in), roughness * roughness, r0, r1));
            s.dir_out = normalize(-dir_in + hm * (2.0f * dot(dir_in, hm)));
            s.roughness = roughness;
        } else s.dir_out = to_world(frame, sample_cos_hemisphere(r0, r1));
        s.valid = true;
        return s;
    }
    return s;
}

LJ_HD bool material_supported(int kind) { return kind == 0 || kind == 1; }

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
    const uint32_t p = sample_id / pass.spp, s = sample_id - p * pass.spp;
    const uint32_t pixel = pass.pixel_list[p];
    const int x = (int)(pixel % (uint32_t)sc.cam.width), y = (int)(pixel / (uint32_t)sc.cam.width);
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
LJ_HD bool shade_path(const DScene &sc, const DPass &pass, PathState &ps, ShadeCounters &cnt) {
    const uint64_t inc = pcg32_inc((uint64_t)pass.pixel_list[ps.sample / pass.spp] * pass.spp + (ps.sample % pass.spp));
    // pending next-event estimation of the previous vertex (path_tracing.h:207)
    if (ps.hcode & HIT_VIS_BIT) ps.rad = ps.rad + ps.nee;
    const bool primary = ps.p2 < 0.0f;
    if (ps.flags & PF_NO_EXT) return false;  // sample_bsdf failed or p2 <= 0 (path_tracing.h:220-223,253-256)
    const int gprim = (ps.hcode & 0x3fffffff) - 1;
    const uint32_t nv_prev = ps.flags & 0xffffu;
    if (gprim < 0) {  // miss: environment map (path_tracing.h:17-27 camera rays, :284-302 bounces)
        if (sc.envmap_light_id >= 0) {
            const DLight &E = sc.lights[sc.envmap_light_id];
            f3 L = light_emission(sc, E, -ps.dir, mk3(0, 0, 0));
            if (primary) ps.rad = ps.rad + L;
            else {
                float p1 = E.pmf * pdf_point_on_light(sc, E, mk3(0, 0, 0), -ps.dir, ps.org);
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
        f3 L = light_emission(sc, EL, -ps.dir, vx.gn);
        if (primary) ps.rad = ps.rad + L;
        else {
            f3 dv = vx.position - ps.org;
            float G = fabsf(dot(ps.dir, vx.gn)) / dot(dv, dv);
            float p2 = ps.p2 * G;
            float p1 = EL.pmf * pdf_point_on_light(sc, EL, vx.position, vx.gn, ps.org);
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
    LightSample pl = sample_point_on_light(sc, Lt, vx.position, lu0, lu1, shape_w);
    ps.nee = mk3(0, 0, 0); ps.sdir = mk3(0, 0, 0); ps.stfar = 0.0f;
    {
        float G; f3 dir_light; float tfar;
        if (Lt.kind == 0) {
            f3 dl = pl.position - vx.position;
            float d2 = dot(dl, dl), d = sqrtf(d2);
            dir_light = normalize(dl);
            tfar = (1.0f - sc.eps) * d;
            G = fmaxf(-dot(dir_light, pl.normal), 0.0f) / d2;
        } else { dir_light = -pl.normal; tfar = INFINITY; G = 1.0f; }
        float p1 = Lt.pmf * pdf_point_on_light(sc, Lt, pl.position, pl.normal, vx.position);
        if (G > 0.0f && p1 > 0.0f) {
            f3 f; float p2;
            bsdf_eval_pdf(sc, mat, dir_view, dir_light, vx, f, p2);
            f3 Le = light_emission(sc, Lt, -dir_light, pl.normal);
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
    BsdfSample bs = bsdf_sample(sc, mat, dir_view, vx, b0, b1, bw);
    ps.org = vx.position;
    ps.flags = nv;
    if (!bs.valid) { ps.flags |= PF_NO_EXT; ps.dir = mk3(0, 0, 1); return ps.stfar > 0.0f; }
    if (bs.eta == 0.0f) ps.spread = fmaxf(ps.spread * (1.0f - bs.roughness) + 0.2f * bs.roughness, 0.0f);  // ray.h:45-51, radius == 0
    else { ps.spread = fmaxf((ps.spread / bs.eta) * (1.0f - bs.roughness) + 0.2f * bs.roughness, 0.0f); ps.eta_scale /= (bs.eta * bs.eta); }
    f3 f; float p2;
    bsdf_eval_pdf(sc, mat, dir_view, bs.dir_out, vx, f, p2);
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
