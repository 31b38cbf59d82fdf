// Participating media on the device: Medium / PhaseFunction / Volume queries (medium.cpp:27-37, media/*.inl,
// phase_functions/*.inl, volume.h:39-144) and the final volumetric path tracer vol_path_tracing with its
// next_event_estimation_final (vol_path_tracing.h:149-163, 299-494, 503-869) — SURVEY row a31, "build last".
//
// Unlike path_tracing, this integrator is not cut into wavefront kernels: a sample walks its whole path in one lane
// (k_volpath), tracing closest hits through the `Tracer` it is handed.  The control flow — free-flight sampling with
// null collisions, shadow connections that pass through index-matched surfaces, medium bookkeeping — is a chain of
// data-dependent loops with several ray casts per bounce; parity with the reference is statistical only, and
// no BASELINE config uses it, so it is built for correctness, not for the roofline.
//
// Tracer provides:  bool closest(f3 org, f3 dir, float tnear, float tfar, float &t, float &u, float &v, int &gprim);
//                   void tick(int slot);   (developer instrumentation of the device kernel: a no-op everywhere else)
#pragma once
#include "dshade.h"

namespace ljd {

LJ_HD f3 vexp3(f3 a) { return mk3(expf(a.x), expf(a.y), expf(a.z)); }
// exp(-majorant * t) with its largest channel scaled to one.  Every quantity a free-flight loop accumulates (the
// transmittance and the two pdfs) carries the same product of these factors and only ever enters the estimator through
// ratios of them, so a common scalar cancels exactly — but unscaled, a long chain of null collisions underflows float
// to 0 / 0 (the reference's doubles have 270 more decades).
LJ_HD f3 vexp3_scaled(f3 a) {
    const float m = fmaxf(fmaxf(a.x, a.y), a.z);
    return mk3(expf(a.x - m), expf(a.y - m), expf(a.z - m));
}
LJ_HD f3 vdiv3(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
LJ_HD float vget3(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
LJ_HD float vavg3(f3 a) { return (a.x + a.y + a.z) / 3.0f; }

// lookup(VolumeSpectrum, p) (volume.h:39-81)
LJ_HD f3 volume_lookup(const DScene &sc, const DVolume &v, f3 p) {
    if (v.kind == 0) return ld3(v.value);
    f3 pn = vdiv3(p - ld3(v.p_min), ld3(v.p_max) - ld3(v.p_min));
    if (pn.x < 0.0f || pn.x > 1.0f || pn.y < 0.0f || pn.y > 1.0f || pn.z < 0.0f || pn.z > 1.0f) return mk3(0, 0, 0);
    const int rx = v.res[0], ry = v.res[1], rz = v.res[2];
    pn.x *= (float)(rx - 1); pn.y *= (float)(ry - 1); pn.z *= (float)(rz - 1);
    auto clampi = [](int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); };
    const int x0 = clampi((int)pn.x, 0, rx - 1), y0 = clampi((int)pn.y, 0, ry - 1), z0 = clampi((int)pn.z, 0, rz - 1);
    const int x1 = clampi(x0 + 1, 0, rx - 1), y1 = clampi(y0 + 1, 0, ry - 1), z1 = clampi(z0 + 1, 0, rz - 1);
    const float dx = pn.x - x0, dy = pn.y - y0, dz = pn.z - z0;
    const float *base = sc.volume_data + v.offset;
    const float w000 = (1 - dx) * (1 - dy) * (1 - dz), w100 = dx * (1 - dy) * (1 - dz), w010 = (1 - dx) * dy * (1 - dz), w110 = dx * dy * (1 - dz),
                w001 = (1 - dx) * (1 - dy) * dz, w101 = dx * (1 - dy) * dz, w011 = (1 - dx) * dy * dz, w111 = dx * dy * dz;
    if (v.mono) {   // one float per voxel: the same sum, once (each channel of the three-float form computes exactly this)
        auto a1 = [&](int x, int y, int z) { return base[(size_t)(z * ry + y) * rx + x]; };
        const float r = (a1(x0, y0, z0) * w000 + a1(x1, y0, z0) * w100 + a1(x0, y1, z0) * w010 + a1(x1, y1, z0) * w110 +
                         a1(x0, y0, z1) * w001 + a1(x1, y0, z1) * w101 + a1(x0, y1, z1) * w011 + a1(x1, y1, z1) * w111) * v.scale;
        return mk3(r, r, r);
    }
    auto at = [&](int x, int y, int z) { return ld3(base + 3 * ((size_t)(z * ry + y) * rx + x)); };
    return (at(x0, y0, z0) * w000 + at(x1, y0, z0) * w100 + at(x0, y1, z0) * w010 + at(x1, y1, z0) * w110 +
            at(x0, y0, z1) * w001 + at(x1, y0, z1) * w101 + at(x0, y1, z1) * w011 + at(x1, y1, z1) * w111) * v.scale;
}
// intersect(Volume, ray) (volume.h:118-144)
LJ_HD bool volume_intersect(const DVolume &v, f3 org, f3 dir, float tfar) {
    if (v.kind == 0) return true;
    float t0 = 0.0f, t1 = tfar;
    const float o[3] = {org.x, org.y, org.z}, d[3] = {dir.x, dir.y, dir.z};
    for (int i = 0; i < 3; i++) {
        float tn = (v.p_min[i] - o[i]) / d[i], tf = (v.p_max[i] - o[i]) / d[i];
        if (tn > tf) { const float s = tn; tn = tf; tf = s; }
        t0 = tn > t0 ? tn : t0; t1 = tf < t1 ? tf : t1;
        if (t0 > t1) return false;
    }
    return true;
}
LJ_HD f3 get_majorant(const DMedium &m, f3 org, f3 dir, float tfar) {
    if (m.kind == 0) return ld3(m.sigma_a) + ld3(m.sigma_s);
    if (!volume_intersect(m.density, org, dir, tfar)) return mk3(0, 0, 0);
    return m.density.kind == 0 ? ld3(m.density.value) : ld3(m.density.max_data) * m.density.scale;
}
LJ_HD void get_sigmas(const DScene &sc, const DMedium &m, f3 p, f3 &sigma_s, f3 &sigma_a) {
    if (m.kind == 0) { sigma_s = ld3(m.sigma_s); sigma_a = ld3(m.sigma_a); return; }
    const f3 density = volume_lookup(sc, m.density, p), albedo = volume_lookup(sc, m.albedo, p);
    sigma_s = density * albedo; sigma_a = density * (mk3(1, 1, 1) - albedo);
}
// phase_functions/*.inl: eval == pdf_sample_phase
LJ_HD float phase_eval(const DMedium &m, f3 dir_in, f3 dir_out) {
    const float inv4pi = 0.25f * kInvPi;
    if (m.phase_kind == 0) return inv4pi;
    const float g = m.g;
    const float b = 1.0f + g * g + 2.0f * g * dot(dir_in, dir_out);
    return inv4pi * (1.0f - g * g) / (b * sqrtf(b));
}
LJ_HD f3 phase_sample(const DMedium &m, f3 dir_in, float r0, float r1) {
    const float g = m.g;
    if (m.phase_kind == 0 || fabsf(g) < 1e-3f) {
        const float z = 1.0f - 2.0f * r0, r = sqrtf(fmaxf(0.0f, 1.0f - z * z)), phi = kTwoPi * r1;
        return mk3(r * cosf(phi), r * sinf(phi), z);
    }
    const float tmp = (g * g - 1.0f) / (2.0f * r0 * g - (g + 1.0f));
    const float cos_el = (tmp * tmp - (1.0f + g * g)) / (2.0f * g);
    const float sin_el = sqrtf(fmaxf(1.0f - cos_el * cos_el, 0.0f));
    const float az = kTwoPi * r1;
    return to_world(make_frame(dir_in), mk3(sin_el * cosf(az), sin_el * sinf(az), cos_el));
}

// update_medium (vol_path_tracing.h:149-163)
LJ_HD int update_medium(const DScene &sc, const DVertex &vx, f3 dir, int medium) {
    const int shape = sc.prims[vx.gprim].shape_id;
    const int interior = sc.shape_media[2 * shape], exterior = sc.shape_media[2 * shape + 1];
    if (interior != exterior) medium = dot(dir, vx.gn) > 0.0f ? exterior : interior;
    return medium;
}

struct VolRng { uint64_t state, inc; };
LJ_HD float vrnd(VolRng &r) { return pcg32_real(r.state, r.inc); }

// next_event_estimation_final (vol_path_tracing.h:299-494)
// Ft: the scene's feature set (dshade.h) — the kernel is instantiated per set, so that a scene of diffuse surfaces does not carry nine BSDFs
template <class Ft, class Tracer>
LJ_HD f3 vol_nee(const DScene &sc, Tracer &tr, VolRng &rng, f3 p, int current_medium, int bounces, f3 dir_view, bool is_surface, const DVertex &vertex) {
    const float lu0 = vrnd(rng), lu1 = vrnd(rng), light_w = vrnd(rng), shape_w = vrnd(rng);
    const int light_id = sample_cdf(sc.light_cdf, sc.n_lights, light_w);
    const DLight &Lt = sc.lights[light_id];
    const LightSample pl = sample_point_on_light<Ft>(sc, Lt, p, lu0, lu1, shape_w);
    // direction and distances from the double sample point (LightSample, dshade.h): exact on a sphere light
    auto dist_to = [&](f3 from) {
        const double dx = pl.dpos[0] - (double)from.x, dy = pl.dpos[1] - (double)from.y, dz = pl.dpos[2] - (double)from.z;
        return sqrt(dx * dx + dy * dy + dz * dz);
    };
    f3 dir_light;
    {
        const double dd = dist_to(p);
        dir_light = mk3((float)((pl.dpos[0] - (double)p.x) / dd), (float)((pl.dpos[1] - (double)p.y) / dd), (float)((pl.dpos[2] - (double)p.z) / dd));
    }
    const f3 p_prime = pl.position, p_origin = p;
    int shadow_medium = current_medium, shadow_bounces = 0;
    f3 T = mk3(1, 1, 1), p_trans_nee = mk3(1, 1, 1), p_trans_dir = mk3(1, 1, 1);
    // (every pass moves p at least eps along the ray, so the walk ends; the cap only bounds a degenerate scene)
    for (int segment = 0;; segment++) {
        if (segment >= 4096) return mk3(0, 0, 0);
        tr.tick(5);
        const double dist_d = dist_to(p);
        const float dist_to_light = (float)dist_d;
        float t, hu, hv; int gprim;
        const bool hit = tr.closest(p, dir_light, sc.eps, (float)((1.0 - (double)sc.eps) * dist_d), t, hu, hv, gprim);
        DVertex sv;
        float next_t = dist_to_light;
        if (hit) { sv = build_vertex(sc, p, dir_light, t, hu, hv, gprim, 0.0f); next_t = length(sv.position - p); }
        if (shadow_medium != -1) {
            const DMedium &med = sc.media[shadow_medium];
            const f3 majorant = get_majorant(med, p, dir_light, (1.0f - sc.eps) * dist_to_light);
            const float u = vrnd(rng);
            const int channel = (int)(u * 3.0f) < 0 ? 0 : ((int)(u * 3.0f) > 2 ? 2 : (int)(u * 3.0f));
            float accum_t = 0.0f; int iteration = 0;
            for (;;) {
                if (vget3(majorant, channel) <= 0.0f) break;
                if (iteration >= sc.max_null_collisions) break;
                tr.tick(4);
                const float tt = -logf(1.0f - vrnd(rng)) / vget3(majorant, channel);
                const float dt = next_t - accum_t;
                accum_t = fminf(accum_t + tt, next_t);
                if (tt < dt) {
                    f3 ss, sa; get_sigmas(sc, med, p + dir_light * accum_t, ss, sa);
                    const f3 ratio = vdiv3(ss + sa, majorant), one_minus = mk3(1, 1, 1) - ratio;
                    const f3 e = vexp3_scaled(-(majorant * tt));
                    const float mx = max3(majorant);
                    T = T * (e * (majorant * one_minus) / mx);
                    p_trans_nee = p_trans_nee * (e * majorant / mx);
                    p_trans_dir = p_trans_dir * (e * majorant * one_minus / mx);
                    if (max3(T) <= 0.0f) break;
                } else {
                    const f3 e = vexp3_scaled(-(majorant * dt));
                    T = T * e; p_trans_nee = p_trans_nee * e; p_trans_dir = p_trans_dir * e;
                    break;
                }
                iteration++;
            }
        }
        if (!hit) break;
        if (sv.material_id >= 0) return mk3(0, 0, 0);
        shadow_bounces++;
        if (sc.max_depth != -1 && bounces + shadow_bounces >= sc.max_depth) return mk3(0, 0, 0);
        shadow_medium = update_medium(sc, sv, dir_light, shadow_medium);
        p = p + dir_light * next_t;
    }
    if (!(max3(T) > 0.0f)) return mk3(0, 0, 0);
    const f3 Le = light_emission<Ft>(sc, Lt, -dir_light, pl.normal);
    const f3 dpl = p_origin - p_prime;
    const float jacobian = fmaxf(-dot(dir_light, pl.normal), 0.0f) / dot(dpl, dpl);
    const f3 pdf_nee = p_trans_nee * (Lt.pmf * pdf_point_on_light<Ft>(sc, Lt, pl.position, pl.normal, p_origin));
    f3 f, pdf_dir;
    if (is_surface) {
        float pdf_bsdf;
        bsdf_eval_pdf<Ft>(sc, sc.materials[vertex.material_id], dir_view, dir_light, vertex, f, pdf_bsdf);
        if (pdf_bsdf <= 0.0f) return mk3(0, 0, 0);
        pdf_dir = p_trans_dir * (pdf_bsdf * jacobian);
    } else {
        const DMedium &med = sc.media[current_medium];
        (void)vrnd(rng); (void)vrnd(rng);   // phase_uv, drawn and never used (vol_path_tracing.h:475)
        const float ph = phase_eval(med, dir_view, dir_light);
        f = mk3(ph, ph, ph);
        pdf_dir = p_trans_dir * (ph * jacobian);
    }
    const f3 contrib = T * f * Le * (jacobian / vavg3(pdf_nee));
    const f3 n2 = pdf_nee * pdf_nee, d2 = pdf_dir * pdf_dir;
    return contrib * vdiv3(n2, n2 + d2);
}

// vol_path_tracing_1 (vol_path_tracing.h:6-41): absorption only — a directly visible emitter through the exterior medium of its surface
template <class Ft, class Tracer>
LJ_HD f3 vol_path_sample_1(const DScene &sc, Tracer &tr, int x, int y, VolRng &rng) {
    const float jy = vrnd(rng), jx = vrnd(rng);
    const f3 org = ld3(sc.cam.org), dir = camera_primary_dir(sc.cam, x, y, jx, jy);
    float t, hu, hv; int gprim;
    if (!tr.closest(org, dir, 0.0f, INFINITY, t, hu, hv, gprim)) return mk3(0, 0, 0);   // (sample_primary's tnear: camera.cpp:46)
    const DVertex vertex = build_vertex(sc, org, dir, t, hu, hv, gprim, 0.0f);
    const int exterior = sc.shape_media[2 * sc.prims[gprim].shape_id + 1];
    if (exterior == -1) return mk3(0, 0, 0);
    const float t_hit = length(vertex.position - org);
    f3 sigma_s, sigma_a; get_sigmas(sc, sc.media[exterior], vertex.position, sigma_s, sigma_a);
    const f3 transmittance = vexp3(-(sigma_a * t_hit));
    f3 Le = mk3(0, 0, 0);
    if (vertex.light_id >= 0) Le = light_emission<Ft>(sc, sc.lights[vertex.light_id], -dir, vertex.gn);
    return transmittance * Le;
}

// vol_path_tracing_2 (vol_path_tracing.h:46-147): one monochromatic homogeneous medium, single scattering, free flight on the red channel
// (oracle/lj_oracle.cpp vol_path_tracing_2 lists what is kept of the reference's reading of an empty optional)
template <class Ft, class Tracer>
LJ_HD f3 vol_path_sample_2(const DScene &sc, Tracer &tr, int x, int y, VolRng &rng) {
    const float jy = vrnd(rng), jx = vrnd(rng);
    const f3 org = ld3(sc.cam.org), dir = camera_primary_dir(sc.cam, x, y, jx, jy);
    float th, hu, hv; int gprim;
    const bool hit = tr.closest(org, dir, 0.0f, INFINITY, th, hu, hv, gprim);
    DVertex vertex;
    float t_hit = INFINITY;
    int medium_id = sc.cam_medium;
    if (hit) { vertex = build_vertex(sc, org, dir, th, hu, hv, gprim, 0.0f); t_hit = length(vertex.position - org); medium_id = sc.shape_media[2 * sc.prims[gprim].shape_id + 1]; }
    if (medium_id < 0) return mk3(0, 0, 0);
    const DMedium &med = sc.media[medium_id];
    f3 sigma_s, sigma_a; get_sigmas(sc, med, hit ? vertex.position : org, sigma_s, sigma_a);
    const f3 sigma_t = sigma_s + sigma_a;
    const float u = vrnd(rng);
    const float t = -logf(1.0f - u) / sigma_t.x;
    if (t < t_hit) {
        const f3 transmittance = vexp3(-(sigma_t * t)), trans_pdf = transmittance * sigma_t;
        const f3 p = org + dir * t;
        const float lu0 = vrnd(rng), lu1 = vrnd(rng), light_w = vrnd(rng), shape_w = vrnd(rng);
        const int light_id = sample_cdf(sc.light_cdf, sc.n_lights, light_w);
        const DLight &Lt = sc.lights[light_id];
        const LightSample pl = sample_point_on_light<Ft>(sc, Lt, p, lu0, lu1, shape_w);
        // (direction and distance from the double sample point, as in vol_nee)
        const double dx = pl.dpos[0] - (double)p.x, dy = pl.dpos[1] - (double)p.y, dz = pl.dpos[2] - (double)p.z;
        const double dist_d = sqrt(dx * dx + dy * dy + dz * dz);
        const f3 dir_light = mk3((float)(dx / dist_d), (float)(dy / dist_d), (float)(dz / dist_d));
        const float dist = (float)dist_d;
        const float rho = phase_eval(med, -dir, dir_light);
        const f3 Le = light_emission<Ft>(sc, Lt, -dir_light, pl.normal);
        const f3 exp_term = vexp3(-(sigma_t * dist));
        float st, su, sv; int sg;
        const float visibility = tr.closest(p, dir_light, sc.eps, (float)((1.0 - (double)sc.eps) * dist_d), st, su, sv, sg) ? 0.0f : 1.0f;
        const float jacobian = fabsf(dot(dir_light, pl.normal)) / (float)(dist_d * dist_d) * visibility;
        const f3 L_s1 = Le * rho * exp_term * jacobian;
        const float L_s1_pdf = Lt.pmf * pdf_point_on_light<Ft>(sc, Lt, pl.position, pl.normal, p);
        return vdiv3(transmittance, trans_pdf) * sigma_s * (L_s1 / L_s1_pdf);
    }
    // a ray that left the scene has no vertex to take an emission from (this arm is reached for it when sigma_t.x is 0 at the ray origin:
    // t is then inf, or nan for u = 0, and `t < inf` fails)
    if (!hit) return mk3(0, 0, 0);
    const f3 transmittance = vexp3(-(sigma_t * t_hit));   // (the pdf is the same expression: the ratio is 1, or 0 / 0 once it underflows)
    f3 Le = mk3(0, 0, 0);
    if (vertex.light_id >= 0) Le = light_emission<Ft>(sc, sc.lights[vertex.light_id], -dir, vertex.gn);
    return vdiv3(transmittance, transmittance) * Le;
}

// vol_path_tracing (vol_path_tracing.h:503-869); see oracle/lj_oracle.cpp for the list of reference quirks kept.  Versions 3, 4 and 5 of
// the reference return this function's result in their first statement (vol_path_tracing.h:880, 1052, 1297).
// Cut at the top of its bounce loop: vol_path_begin sets up a camera sample, vol_path_step runs ONE iteration of the loop at
// vol_path_tracing.h:524 and says whether there is another.  k_volpath keeps a path per lane and gives a lane whose path has ended the
// next camera sample (regeneration), so a wave's lanes stay busy whatever their paths' lengths; vol_path_sample below is the two
// chained, operation for operation what it was as one function.
struct VolPath {
    VolRng rng;
    f3 org, dir; float spread;          // RayDifferential{0, 0}: only `spread` ever changes (ray.h:45-66 with radius 0)
    int current_medium;
    f3 throughput, radiance;
    int bounces;
    float dir_pdf; f3 nee_p_cache, multi_trans_pdf;
    float eta_scale;
    uint32_t bounce_iterations;         // (statistics: iterations that reached the scattering / shading part)
    int guard;                          // Russian roulette ends a path with probability >= 5 % per iteration past rr_depth; the cap bounds rr_depth = huge
};
// returns false when the sample is already finished (`result`): versions 1 and 2 are single-shot estimators
template <class Ft, class Tracer>
LJ_HD bool vol_path_begin(const DScene &sc, Tracer &tr, int x, int y, uint64_t stream, uint64_t seed, VolPath &P, f3 &result) {
    P.rng.inc = pcg32_inc(stream); P.rng.state = pcg32_init(stream, seed);
    P.bounce_iterations = 0; P.guard = 0;
    if (sc.vol_path_version == 1) { result = vol_path_sample_1<Ft>(sc, tr, x, y, P.rng); return false; }
    if (sc.vol_path_version == 2) { result = vol_path_sample_2<Ft>(sc, tr, x, y, P.rng); return false; }
    const float jy = vrnd(P.rng), jx = vrnd(P.rng);
    P.org = ld3(sc.cam.org); P.dir = camera_primary_dir(sc.cam, x, y, jx, jy);
    P.spread = 0.0f;
    P.current_medium = sc.cam_medium;
    P.throughput = mk3(1, 1, 1); P.radiance = mk3(0, 0, 0);
    P.bounces = 0;
    P.dir_pdf = 0.0f; P.nee_p_cache = mk3(0, 0, 0);
    P.multi_trans_pdf = mk3(1, 1, 1);
    P.eta_scale = 1.0f;
    return true;
}
// one iteration of the loop; false: the path has ended with `result`
template <class Ft, class Tracer>
LJ_HD bool vol_path_step(const DScene &sc, Tracer &tr, VolPath &P, f3 &result) {
    VolRng &rng = P.rng;
    f3 &org = P.org, &dir = P.dir; float &spread = P.spread;
    int &current_medium = P.current_medium;
    f3 &throughput = P.throughput, &radiance = P.radiance;
    int &bounces = P.bounces;
    float &dir_pdf = P.dir_pdf; f3 &nee_p_cache = P.nee_p_cache, &multi_trans_pdf = P.multi_trans_pdf;
    float &eta_scale = P.eta_scale;
    uint32_t &bounces_out = P.bounce_iterations;
    if (P.guard >= 65536) { result = radiance; return false; }
    P.guard++;
    bool scatter = false;
    float t, hu, hv; int gprim;
    const bool hit = tr.closest(org, dir, sc.eps, INFINITY, t, hu, hv, gprim);
    DVertex vertex;
    float t_hit = INFINITY;
    if (hit) { vertex = build_vertex(sc, org, dir, t, hu, hv, gprim, spread); t_hit = length(vertex.position - org); }
    f3 transmittance = mk3(1, 1, 1), trans_dir_pdf = mk3(1, 1, 1), trans_nee_pdf = mk3(1, 1, 1);
    if (current_medium != -1) {
        const DMedium &med = sc.media[current_medium];
        const f3 majorant = get_majorant(med, org, dir, INFINITY);
        const float u = vrnd(rng);
        const int channel = (int)(u * 3.0f) < 0 ? 0 : ((int)(u * 3.0f) > 2 ? 2 : (int)(u * 3.0f));
        float accum_t = 0.0f; int iteration = 0;
        for (;;) {
            if (vget3(majorant, channel) <= 0.0f) break;
            if (iteration >= sc.max_null_collisions) break;
            tr.tick(3);
            const float tt = -logf(1.0f - vrnd(rng)) / vget3(majorant, channel);
            const float dt = t_hit - accum_t;
            accum_t = fminf(accum_t + tt, t_hit);
            if (tt < dt) {
                const f3 p = org + dir * accum_t;
                f3 ss, sa; get_sigmas(sc, med, p, ss, sa);
                const f3 real_prob = vdiv3(ss + sa, majorant), one_minus = mk3(1, 1, 1) - real_prob;
                const f3 e = vexp3_scaled(-(majorant * tt));
                const float mx = max3(majorant);
                if (vrnd(rng) < vget3(real_prob, channel)) {
                    scatter = true;
                    transmittance = transmittance * (e / mx);
                    trans_dir_pdf = trans_dir_pdf * (e * majorant * real_prob / mx);
                    org = p;
                    break;
                }
                transmittance = transmittance * (e * (majorant * one_minus) / mx);
                trans_dir_pdf = trans_dir_pdf * (e * majorant * one_minus / mx);
                trans_nee_pdf = trans_nee_pdf * (e * majorant / mx);
            } else {
                const f3 e = vexp3_scaled(-(majorant * dt));
                transmittance = transmittance * e; trans_dir_pdf = trans_dir_pdf * e; trans_nee_pdf = trans_nee_pdf * e;
                org = vertex.position;
                break;
            }
            iteration++;
        }
        multi_trans_pdf = multi_trans_pdf * trans_dir_pdf;
    } else {
        if (hit) org = vertex.position;
        else { result = mk3(0, 0, 0); return false; }
    }
    throughput = throughput * (transmittance / vavg3(trans_dir_pdf));
    if (!scatter && hit && vertex.light_id >= 0) {
        const DLight &EL = sc.lights[vertex.light_id];
        const f3 Le = light_emission<Ft>(sc, EL, -dir, vertex.gn);
        if (bounces == 0) { result = radiance + throughput * Le; return false; }
        const f3 pdf_nee = trans_nee_pdf * (EL.pmf * pdf_point_on_light<Ft>(sc, EL, vertex.position, vertex.gn, nee_p_cache));
        const f3 dv = nee_p_cache - vertex.position;
        const float jacobian = fmaxf(-dot(-dir, vertex.gn), 0.0f) / dot(dv, dv);
        const f3 pdf_phase = multi_trans_pdf * (dir_pdf * jacobian);
        const f3 p2 = pdf_phase * pdf_phase, n2 = pdf_nee * pdf_nee;
        radiance = radiance + throughput * Le * vdiv3(p2, p2 + n2);
    }
    if (!scatter && hit && vertex.material_id == -1) {
        current_medium = update_medium(sc, vertex, dir, current_medium);
        org = vertex.position;
        bounces++;
        return true;
    }
    if (bounces >= sc.max_depth - 1 && sc.max_depth != -1) { result = radiance; return false; }
    bounces_out++;
    if (scatter && current_medium != -1) {
        const DMedium &med = sc.media[current_medium];
        f3 sigma_s, sigma_a; get_sigmas(sc, med, org, sigma_s, sigma_a);
        const f3 nee = vol_nee<Ft>(sc, tr, rng, org, current_medium, bounces, -dir, false, vertex);
        radiance = radiance + throughput * sigma_s * nee;
        if (max3(nee) > 0.0f) nee_p_cache = org;
        const float r0 = vrnd(rng), r1 = vrnd(rng);
        const f3 next_dir = phase_sample(med, -dir, r0, r1);
        const float phase_pdf = phase_eval(med, -dir, next_dir);
        throughput = throughput * sigma_s * (phase_pdf / phase_pdf);
        dir = next_dir;
        dir_pdf = phase_pdf;
        multi_trans_pdf = mk3(1, 1, 1);
    } else if (hit) {
        const f3 nee = vol_nee<Ft>(sc, tr, rng, org, current_medium, bounces, -dir, true, vertex);
        radiance = radiance + throughput * nee;
        if (max3(nee) > 0.0f) nee_p_cache = org;
        const DMaterial &mat = sc.materials[vertex.material_id];
        const f3 dir_view = -dir;
        const float b0 = vrnd(rng), b1 = vrnd(rng), bw = vrnd(rng);
        const BsdfSample bs = bsdf_sample<Ft>(sc, mat, dir_view, vertex, b0, b1, bw);
        if (!bs.valid) { result = radiance; return false; }
        dir = bs.dir_out;
        if (bs.eta == 0.0f) spread = fmaxf(spread * (1.0f - bs.roughness) + 0.2f * bs.roughness, 0.0f);
        else {
            spread = fmaxf((spread / bs.eta) * (1.0f - bs.roughness) + 0.2f * bs.roughness, 0.0f);
            eta_scale /= (bs.eta * bs.eta);
            current_medium = update_medium(sc, vertex, dir, current_medium);
        }
        f3 f; float pdf_bsdf;
        bsdf_eval_pdf<Ft>(sc, mat, dir_view, bs.dir_out, vertex, f, pdf_bsdf);
        throughput = throughput * (f / pdf_bsdf);
    }
    if (bounces >= sc.rr_depth) {
        const float rr_prob = fminf(max3(throughput * (1.0f / eta_scale)), 0.95f);
        if (vrnd(rng) > rr_prob) { result = radiance; return false; }
        throughput = throughput / rr_prob;
    }
    bounces++;
    return true;
}
template <class Ft = FeatAll, class Tracer>
LJ_HD f3 vol_path_sample(const DScene &sc, Tracer &tr, int x, int y, uint64_t stream, uint64_t seed, uint32_t &bounces_out) {
    VolPath P; f3 result = mk3(0, 0, 0);
    bounces_out = 0;
    if (!vol_path_begin<Ft>(sc, tr, x, y, stream, seed, P, result)) return result;
    while (vol_path_step<Ft>(sc, tr, P, result)) {}
    bounces_out = P.bounce_iterations;
    return result;
}

} // namespace ljd
