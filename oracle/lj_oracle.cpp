// TEST INFRASTRUCTURE — CPU restatement ("oracle") of lajolla's per-pixel-sample hot path.
//
// NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// liblj_oracle.so.  The product (liblajolla_hip.so) never links, loads or calls anything in this directory.
//
// What it is: a double-precision, scalar C++ restatement of the reference algorithm, function by function,
// each citing the reference file:line it follows.  Inputs arrive as the public LjSceneDesc
// (include/lajolla_hip.h), i.e. the constructor arguments of the reference's Scene.
//
// Parity status:
//   PINNED  by tests/golden/*.json, produced by the reference's own functions compiled from /root/reference
//           (oracle/ref_build.sh, oracle/gen_golden.cpp): pcg32, filters, camera rays, frames, table
//           distributions, ray differentials, light selection / sampling / pdf / emission, triangle + sphere
//           shading info (PathVertex), eval / pdf / sample of all nine Material alternatives, scene tables.
//   UNPINNED against the reference: the ray/scene intersection itself (Embree 3.13.2 is a binary-only
//           dependency whose Linux library is absent from /root/reference) and therefore path_tracing()
//           end-to-end.  The bounce loop below is a line-by-line restatement of path_tracing.h:7-325 over
//           pinned callees; the closest-hit / any-hit queries are our own float Plücker test + exact
//           tie-breaking, pinned only by the reference's single intersection fixture
//           (src/tests/intersection.cpp:28-37) and by BVH == brute-force equivalence.
//
// Build: g++ -O2 -ffp-contract=off (no FMA contraction: the float intersection arithmetic below is mirrored
// bit-for-bit by the HIP kernels, which compile it with contraction off as well).
#include "../include/lajolla_hip.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

typedef double Real;
const Real c_PI = Real(3.14159265358979323846);
const Real c_INVFOURPI = Real(1.0) / (Real(4.0) * c_PI);  // lajolla.h
const Real c_INVPI = Real(1.0) / c_PI;
const Real c_TWOPI = Real(2.0) * c_PI;
const Real c_INVTWOPI = Real(1.0) / c_TWOPI;

// ------------------------------------------------------------------ vector.h
struct Vector2 { Real x, y; Real operator[](int i) const { return i == 0 ? x : y; } };
struct Vector3 {
    Real x, y, z;
    Real operator[](int i) const { return (&x)[i]; }
    Real &operator[](int i) { return (&x)[i]; }
};
inline Vector3 operator+(const Vector3 &a, const Vector3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vector3 operator-(const Vector3 &a, const Vector3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vector3 operator-(const Vector3 &a) { return {-a.x, -a.y, -a.z}; }
inline Vector3 operator*(const Vector3 &a, Real s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vector3 operator*(Real s, const Vector3 &a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vector3 operator*(const Vector3 &a, const Vector3 &b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vector3 operator/(const Vector3 &a, Real s) { Real inv = Real(1) / s; return {a.x * inv, a.y * inv, a.z * inv}; }  // vector.h:194-197
inline Vector3 &operator+=(Vector3 &a, const Vector3 &b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline Real dot(const Vector3 &a, const Vector3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vector3 cross(const Vector3 &a, const Vector3 &b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Real length(const Vector3 &a) { return std::sqrt(dot(a, a)); }
inline Real distance_squared(const Vector3 &a, const Vector3 &b) { return dot(a - b, a - b); }
inline Real distance(const Vector3 &a, const Vector3 &b) { return std::sqrt(distance_squared(a, b)); }
inline Vector3 normalize(const Vector3 &a) { Real l = length(a); if (l <= 0) return {0, 0, 0}; return a / l; }  // vector.h:249-257
inline Real vmax(const Vector3 &v) { return std::max(std::max(v.x, v.y), v.z); }
typedef Vector3 Spectrum;
inline Real luminance(const Spectrum &s) { return s.x * Real(0.212671) + s.y * Real(0.715160) + s.z * Real(0.072169); }  // spectrum.h:32-34
inline Real clampr(Real v, Real lo, Real hi) { return v < lo ? lo : (hi < v ? hi : v); }  // std::clamp
inline Real modulo(Real a, Real b) { Real r = std::fmod(a, b); return (r < 0) ? r + b : r; }  // lajolla.h:58-61
inline int modulo(int a, int b) { int r = a % b; return (r < 0) ? r + b : r; }

// ------------------------------------------------------------------ matrix.h / transform.cpp (row-major 4x4)
struct Matrix4x4 { Real m[16]; Real operator()(int i, int j) const { return m[i * 4 + j]; } };
inline Vector3 xform_point(const Matrix4x4 &M, const Vector3 &p) {  // transform.cpp:80-88
    Real x = M(0, 0) * p.x + M(0, 1) * p.y + M(0, 2) * p.z + M(0, 3);
    Real y = M(1, 0) * p.x + M(1, 1) * p.y + M(1, 2) * p.z + M(1, 3);
    Real z = M(2, 0) * p.x + M(2, 1) * p.y + M(2, 2) * p.z + M(2, 3);
    Real w = M(3, 0) * p.x + M(3, 1) * p.y + M(3, 2) * p.z + M(3, 3);
    Real inv_w = Real(1) / w;
    return {x * inv_w, y * inv_w, z * inv_w};
}
inline Vector3 xform_vector(const Matrix4x4 &M, const Vector3 &v) {  // transform.cpp:90-94
    return {M(0, 0) * v.x + M(0, 1) * v.y + M(0, 2) * v.z, M(1, 0) * v.x + M(1, 1) * v.y + M(1, 2) * v.z,
            M(2, 0) * v.x + M(2, 1) * v.y + M(2, 2) * v.z};
}

// ------------------------------------------------------------------ frame.h
struct Frame { Vector3 x, y, n; };
inline void coordinate_system(const Vector3 &n, Vector3 &a_out, Vector3 &b_out) {  // frame.h:11-22
    if (n.z < Real(-1 + 1e-6)) { a_out = {0, -1, 0}; b_out = {-1, 0, 0}; }
    else {
        Real a = 1 / (1 + n.z), b = -n.x * n.y * a;
        a_out = {1 - n.x * n.x * a, b, -n.x}; b_out = {b, 1 - n.y * n.y * a, -n.y};
    }
}
inline Frame make_frame(const Vector3 &n) { Frame f; f.n = n; coordinate_system(n, f.x, f.y); return f; }
inline Frame operator-(const Frame &f) { return {-f.x, -f.y, -f.n}; }
inline Vector3 to_local(const Frame &f, const Vector3 &v) { return {dot(v, f.x), dot(v, f.y), dot(v, f.n)}; }
inline Vector3 to_world(const Frame &f, const Vector3 &v) { return f.x * v.x + f.y * v.y + f.n * v.z; }

// ------------------------------------------------------------------ pcg.h
struct pcg32_state { uint64_t state, inc; };
inline uint32_t next_pcg32(pcg32_state &rng) {  // pcg.h:22-30
    uint64_t oldstate = rng.state;
    rng.state = oldstate * 6364136223846793005ULL + (rng.inc | 1);
    uint32_t xorshifted = uint32_t(((oldstate >> 18u) ^ oldstate) >> 27u);
    uint32_t rot = uint32_t(oldstate >> 59u);
    return uint32_t((xorshifted >> rot) | (xorshifted << ((-rot) & 31)));
}
inline pcg32_state init_pcg32(uint64_t stream_id, uint64_t seed) {  // pcg.h:33-41
    pcg32_state s; s.state = 0U; s.inc = (stream_id << 1u) | 1u;
    next_pcg32(s); s.state += seed; next_pcg32(s);
    return s;
}
inline double next_pcg32_real(pcg32_state &rng) {  // pcg.h:61-68
    union { uint64_t u; double d; } x;
    x.u = ((uint64_t)next_pcg32(rng) << 20) | 0x3ff0000000000000ULL;
    return x.d - 1.0;
}
inline float next_pcg32_float(pcg32_state &rng) {  // pcg.h:50-57
    union { uint32_t u; float f; } x;
    x.u = (next_pcg32(rng) >> 9) | 0x3f800000u;
    return x.f - 1.0f;
}

// ------------------------------------------------------------------ ray.h
struct Ray { Vector3 org, dir; Real tnear, tfar; };
struct RayDifferential { Real radius = 0, spread = 0; };
inline Real rd_transfer(const RayDifferential &r, Real dist) { return r.radius + r.spread * dist; }  // ray.h:40-42
inline Real rd_reflect(const RayDifferential &r, Real mean_curvature, Real roughness) {  // ray.h:45-51
    Real spec_spread = r.spread + 2 * mean_curvature * r.radius;
    return std::fmax(spec_spread * (1 - roughness) + Real(0.2) * roughness, Real(0));
}
inline Real rd_refract(const RayDifferential &r, Real mean_curvature, Real eta, Real roughness) {  // ray.h:59-66
    Real spec_spread = (r.spread + 2 * mean_curvature * r.radius) / eta;
    return std::fmax(spec_spread * (1 - roughness) + Real(0.2) * roughness, Real(0));
}

// ------------------------------------------------------------------ table_dist.cpp
struct TableDist1D { std::vector<Real> pmf, cdf; };
TableDist1D make_table_dist_1d(const std::vector<Real> &f) {  // table_dist.cpp:3-25
    TableDist1D t; t.pmf = f; t.cdf.assign(f.size() + 1, 0);
    for (size_t i = 0; i < f.size(); i++) t.cdf[i + 1] = t.cdf[i] + t.pmf[i];
    Real total = t.cdf.back();
    if (total > 0) { for (size_t i = 0; i < f.size(); i++) { t.pmf[i] /= total; t.cdf[i] /= total; } }  // last entry stays = total
    else { for (size_t i = 0; i < f.size(); i++) { t.pmf[i] = Real(1) / Real(f.size()); t.cdf[i] = Real(i) / Real(f.size()); } t.cdf.back() = 1; }
    return t;
}
int sample_1d(const TableDist1D &t, Real u) {  // table_dist.cpp:27-33
    int size = (int)t.pmf.size();
    const Real *ptr = std::upper_bound(t.cdf.data(), t.cdf.data() + size + 1, u);
    int off = (int)(ptr - t.cdf.data() - 1);
    return off < 0 ? 0 : (off > size - 1 ? size - 1 : off);
}
struct TableDist2D { std::vector<Real> cdf_rows, pdf_rows, cdf_marginals, pdf_marginals; Real total_values = 0; int width = 0, height = 0; };
TableDist2D make_table_dist_2d(const std::vector<Real> &f, int width, int height) {  // table_dist.cpp:40-114
    TableDist2D t; t.width = width; t.height = height;
    t.cdf_rows.assign((size_t)height * (width + 1), 0); t.pdf_rows.assign((size_t)height * width, 0);
    for (int y = 0; y < height; y++) {
        Real *cdf = &t.cdf_rows[(size_t)y * (width + 1)];
        cdf[0] = 0;
        for (int x = 0; x < width; x++) cdf[x + 1] = cdf[x] + f[(size_t)y * width + x];
        Real integral = cdf[width];
        if (integral > 0) {
            for (int x = 0; x < width; x++) cdf[x] /= integral;
            for (int x = 0; x < width; x++) t.pdf_rows[(size_t)y * width + x] = f[(size_t)y * width + x] / integral;
        } else {
            for (int x = 0; x < width; x++) { t.pdf_rows[(size_t)y * width + x] = Real(1) / Real(width); cdf[x] = Real(x) / Real(width); }
            cdf[width] = 1;
        }
    }
    t.cdf_marginals.assign(height + 1, 0); t.pdf_marginals.assign(height, 0);
    for (int y = 0; y < height; y++) t.cdf_marginals[y + 1] = t.cdf_marginals[y] + t.cdf_rows[(size_t)y * (width + 1) + width];
    t.total_values = t.cdf_marginals.back();
    if (t.total_values > 0) {
        for (int y = 0; y < height; y++) t.cdf_marginals[y] /= t.total_values;
        t.cdf_marginals[height] = 1;
        for (int y = 0; y < height; y++) t.pdf_marginals[y] = t.cdf_rows[(size_t)y * (width + 1) + width] / t.total_values;
    } else {
        for (int y = 0; y < height; y++) { t.pdf_marginals[y] = Real(1) / Real(height); t.cdf_marginals[y] = Real(y) / Real(height); }
        t.cdf_marginals[height] = 1;
    }
    for (int y = 0; y < height; y++) t.cdf_rows[(size_t)y * (width + 1) + width] = 1;
    return t;
}
Vector2 sample_2d(const TableDist2D &t, const Vector2 &rnd) {  // table_dist.cpp:116-139
    int w = t.width, h = t.height;
    const Real *yp = std::upper_bound(t.cdf_marginals.data(), t.cdf_marginals.data() + h + 1, rnd.y);
    int yo = std::min(std::max((int)(yp - t.cdf_marginals.data() - 1), 0), h - 1);
    Real dy = rnd.y - t.cdf_marginals[yo];
    if ((t.cdf_marginals[yo + 1] - t.cdf_marginals[yo]) > 0) dy /= (t.cdf_marginals[yo + 1] - t.cdf_marginals[yo]);
    const Real *cdf = &t.cdf_rows[(size_t)yo * (w + 1)];
    const Real *xp = std::upper_bound(cdf, cdf + w + 1, rnd.x);
    int xo = std::min(std::max((int)(xp - cdf - 1), 0), w - 1);
    Real dx = rnd.x - cdf[xo];
    if (cdf[xo + 1] - cdf[xo] > 0) dx /= (cdf[xo + 1] - cdf[xo]);
    return {(xo + dx) / w, (yo + dy) / h};
}
Real pdf_2d(const TableDist2D &t, const Vector2 &xy) {  // table_dist.cpp:141-151
    int w = t.width, h = t.height;
    int x = (int)clampr(xy.x * w, Real(0), Real(w - 1));
    int y = (int)clampr(xy.y * h, Real(0), Real(h - 1));
    return t.pdf_marginals[y] * t.pdf_rows[(size_t)y * w + x] * w * h;
}

// ------------------------------------------------------------------ mipmap.h / texture.h
struct Mip { int levels = 0; std::vector<int> w, h; std::vector<std::vector<Vector3>> data; };  // 1-channel images replicate into .x
Mip make_mipmap(const LjImage &img) {  // mipmap.h:25-48
    Mip m;
    std::vector<Vector3> l0((size_t)img.width * img.height);
    for (size_t i = 0; i < l0.size(); i++) {
        if (img.channels >= 3) l0[i] = {img.data[i * img.channels], img.data[i * img.channels + 1], img.data[i * img.channels + 2]};
        else l0[i] = {img.data[i * img.channels], img.data[i * img.channels], img.data[i * img.channels]};
    }
    int size = std::max(img.width, img.height);
    int num_levels = std::min((int)std::ceil(std::log2(Real(size)) + 1), 8);
    m.w.push_back(img.width); m.h.push_back(img.height); m.data.push_back(std::move(l0));
    for (int i = 1; i < num_levels; i++) {
        int pw = m.w.back(), ph = m.h.back();
        int nw = std::max(pw / 2, 1), nh = std::max(ph / 2, 1);
        const std::vector<Vector3> &prev = m.data.back();
        std::vector<Vector3> next((size_t)nw * nh);
        // NB: like the reference, reads prev(2x+1, 2y+1) without clamping; for 1-wide levels that indexes the next
        // row / past the end.  All shipped textures have >= 2 texels on both axes at every level that is halved
        // except the last ones of very elongated images; we clamp to stay in bounds (values there are unpinned).
        auto at = [&](int x, int y) { size_t idx = (size_t)y * pw + x; return prev[std::min(idx, prev.size() - 1)]; };
        for (int y = 0; y < nh; y++) for (int x = 0; x < nw; x++)
            next[(size_t)y * nw + x] = (at(2 * x, 2 * y) + at(2 * x + 1, 2 * y) + at(2 * x, 2 * y + 1) + at(2 * x + 1, 2 * y + 1)) / Real(4);
        m.w.push_back(nw); m.h.push_back(nh); m.data.push_back(std::move(next));
    }
    m.levels = num_levels;
    return m;
}
Vector3 mip_lookup_level(const Mip &m, Real u, Real v, int level) {  // mipmap.h:52-73
    int W = m.w[level], H = m.h[level];
    u = u * W - Real(0.5); v = v * H - Real(0.5);
    int ufi = modulo(int(u), W), vfi = modulo(int(v), H);
    int uci = modulo(ufi + 1, W), vci = modulo(vfi + 1, H);
    Real uo = u - ufi, vo = v - vfi;
    const std::vector<Vector3> &d = m.data[level];
    Vector3 ff = d[(size_t)vfi * W + ufi], fc = d[(size_t)vci * W + ufi], cf = d[(size_t)vfi * W + uci], cc = d[(size_t)vci * W + uci];
    return ff * (1 - uo) * (1 - vo) + fc * (1 - uo) * vo + cf * uo * (1 - vo) + cc * uo * vo;
}
Vector3 mip_lookup(const Mip &m, Real u, Real v, Real level) {  // mipmap.h:76-89
    if (level <= 0) return mip_lookup_level(m, u, v, 0);
    if (level < Real(m.levels - 1)) {
        int fl = std::min(std::max((int)std::floor(level), 0), m.levels - 1);
        int cl = std::min(std::max(fl + 1, 0), m.levels - 1);
        Real lo = level - fl;
        return mip_lookup_level(m, u, v, fl) * (1 - lo) + mip_lookup_level(m, u, v, cl) * lo;
    }
    return mip_lookup_level(m, u, v, m.levels - 1);
}

struct OScene;
Vector3 eval_texture(const OScene *sc, const LjTexture &t, bool spectrum, const Vector2 &uv, Real footprint);

// ------------------------------------------------------------------ intersection.h PathVertex
struct PathVertex {
    Vector3 position, geometry_normal; Frame shading_frame; Vector2 st, uv;
    Real uv_screen_size, mean_curvature, ray_radius;
    int shape_id = -1, primitive_id = -1, material_id = -1;
};
struct PointAndNormal { Vector3 position, normal; };

// ------------------------------------------------------------------ float ray / primitive tests (our definition)
// These replace Embree's rtcIntersect1 / rtcOccluded1 kernels (intersection.cpp:32,83).  Plücker-coordinate
// edge tests on origin-relative float vertices: the value computed for an edge is exactly negated for the
// neighbouring triangle that shares it, so a ray cannot slip between two triangles (the property the reference
// asks Embree for with RTC_SCENE_FLAG_ROBUST, scene.cpp:23).  Every operation is a single IEEE float op (or an
// explicit fused multiply-add) in the written order; the HIP kernels repeat it verbatim.
struct Hit { float t, u, v; int shape_id, prim_id; long long gprim; double t_sphere; };
// a*b - c*d and x*dx + y*dy + z*dz with explicit fused multiply-adds (IEEE fma is exactly specified, so the CPU oracle and
// the GPU agree bit for bit).  Negating (a, c) negates the first exactly and negating (x, y, z) the second, which is what
// keeps the edge tests of two triangles sharing an edge exact mirror images.
#define LJ_TRI_CROSS(a, b, c, d) __builtin_fmaf((a), (b), -((c) * (d)))
#define LJ_TRI_DOT(x, y, z, dx, dy, dz) __builtin_fmaf((z), (dz), __builtin_fmaf((y), (dy), (x) * (dx)))
inline bool tri_test(const float o[3], const float d[3], float tnear, float tfar,
                     const float p0[3], const float p1[3], const float p2[3], float &t_out, float &u_out, float &v_out) {
    float ax = p0[0] - o[0], ay = p0[1] - o[1], az = p0[2] - o[2];
    float bx = p1[0] - o[0], by = p1[1] - o[1], bz = p1[2] - o[2];
    float cx = p2[0] - o[0], cy = p2[1] - o[1], cz = p2[2] - o[2];
    float e0x = cx - ax, e0y = cy - ay, e0z = cz - az;
    float e1x = ax - bx, e1y = ay - by, e1z = az - bz;
    float e2x = bx - cx, e2y = by - cy, e2z = bz - cz;
    float s0x = cx + ax, s0y = cy + ay, s0z = cz + az;
    float s1x = ax + bx, s1y = ay + by, s1z = az + bz;
    float s2x = bx + cx, s2y = by + cy, s2z = bz + cz;
    float U = LJ_TRI_DOT(LJ_TRI_CROSS(e0y, s0z, e0z, s0y), LJ_TRI_CROSS(e0z, s0x, e0x, s0z), LJ_TRI_CROSS(e0x, s0y, e0y, s0x), d[0], d[1], d[2]);
    float V = LJ_TRI_DOT(LJ_TRI_CROSS(e1y, s1z, e1z, s1y), LJ_TRI_CROSS(e1z, s1x, e1x, s1z), LJ_TRI_CROSS(e1x, s1y, e1y, s1x), d[0], d[1], d[2]);
    float W = LJ_TRI_DOT(LJ_TRI_CROSS(e2y, s2z, e2z, s2y), LJ_TRI_CROSS(e2z, s2x, e2x, s2z), LJ_TRI_CROSS(e2x, s2y, e2y, s2x), d[0], d[1], d[2]);
    float mn = fminf(fminf(U, V), W), mx = fmaxf(fmaxf(U, V), W);
    if (!(mn >= 0.0f || mx <= 0.0f)) return false;
    float S = (U + V) + W;
    if (S == 0.0f) return false;
    // Ng = (p1-p0) x (p2-p0) = e1 x e0 with e1 = a-b, e0 = c-a
    float nx = LJ_TRI_CROSS(e1y, e0z, e1z, e0y), ny = LJ_TRI_CROSS(e1z, e0x, e1x, e0z), nz = LJ_TRI_CROSS(e1x, e0y, e1y, e0x);
    float den = LJ_TRI_DOT(nx, ny, nz, d[0], d[1], d[2]);
    if (den == 0.0f) return false;
    float T = LJ_TRI_DOT(nx, ny, nz, ax, ay, az);
    float t = T / den;
    if (!(t > tnear && t <= tfar)) return false;
    float rS = 1.0f / S;
    t_out = t; u_out = U * rS; v_out = V * rS;
    return true;
}
// sphere.inl:15-38 — the reference's own callback arithmetic: double maths on the float ray.
inline bool solve_quadratic(Real a, Real b, Real c, Real *t0, Real *t1) {
    if (a == 0) { if (b == 0) return false; *t0 = *t1 = -c / b; return true; }
    Real disc = b * b - 4 * a * c;
    if (disc < 0) return false;
    Real rd = std::sqrt(disc);
    if (b >= 0) { *t0 = (-b - rd) / (2 * a); *t1 = 2 * c / (-b - rd); }
    else { *t0 = 2 * c / (-b + rd); *t1 = (-b + rd) / (2 * a); }
    return true;
}
inline bool sphere_test(const float o[3], const float d[3], float tnear, float tfar, const Vector3 &center, Real radius, Real &t_out) {  // sphere.inl:40-84,103-141
    Vector3 org{o[0], o[1], o[2]}, dir{d[0], d[1], d[2]};
    Vector3 v = org - center;
    Real A = dot(dir, dir), B = 2 * dot(dir, v), C = dot(v, v) - radius * radius;
    Real t0, t1;
    if (!solve_quadratic(A, B, C, &t0, &t1)) return false;
    Real t = -1;
    if (t0 >= tnear && t0 < tfar) t = t0;
    if (t1 >= tnear && t1 < tfar && t < 0) t = t1;
    if (t >= tnear && t < tfar) { t_out = t; return true; }
    return false;
}

// ------------------------------------------------------------------ scene (scene.h / scene.cpp)
struct OMesh { const double *P, *N, *UV; const int32_t *I; int64_t nv, nt; bool has_n, has_uv; Real total_area = 0; TableDist1D tri_sampler; std::vector<float> Pf; long long gprim0 = 0; };
struct BvhNode { float lo[3], hi[3]; int left, right, first, count, axis; };
struct OScene {
    LjSceneDesc d;
    std::vector<LjShape> shapes; std::vector<LjMaterial> materials; std::vector<LjLight> lights;
    std::vector<OMesh> meshes;  // per shape (empty for spheres)
    std::vector<long long> sphere_gprim;
    std::vector<Mip> mips3, mips1;
    std::vector<TableDist2D> env_dist;  // per light
    TableDist1D light_dist;
    Real bounds_radius = 0; Vector3 bounds_center{0, 0, 0};
    Matrix4x4 sample_to_cam, cam_to_world;
    // primitive list for intersection: (shape, prim) in global order
    std::vector<int> prim_shape, prim_local; std::vector<int> prim_order; std::vector<BvhNode> nodes;
    bool use_bvh = false;
    uint64_t seed = 0x853c49e6748fea9bULL;
};

Real shadow_epsilon(const OScene &s) { return std::min(s.bounds_radius * Real(1e-5), Real(0.01)); }  // scene.h:99-105

Vector3 eval_texture(const OScene *sc, const LjTexture &t, bool spectrum, const Vector2 &uv, Real footprint) {  // texture.h:123-154
    if (t.kind == LJ_TEX_CONSTANT) return {t.value[0], t.value[1], t.value[2]};
    Vector2 local_uv{modulo(uv.x * t.uscale + t.uoffset, Real(1)), modulo(uv.y * t.vscale + t.voffset, Real(1))};
    if (t.kind == LJ_TEX_IMAGE) {
        const Mip &img = spectrum ? sc->mips3[t.texture_id] : sc->mips1[t.texture_id];
        Real scaled_footprint = std::max(img.w[0], img.h[0]) * std::max(t.uscale, t.vscale) * footprint;
        Real level = std::log2(std::max(scaled_footprint, Real(1e-8f)));
        return mip_lookup(img, local_uv.x, local_uv.y, level);
    }
    int x = 2 * modulo((int)(local_uv.x * 2), 2) - 1, y = 2 * modulo((int)(local_uv.y * 2), 2) - 1;
    if (x * y == 1) return {t.value[0], t.value[1], t.value[2]};
    return {t.color1[0], t.color1[1], t.color1[2]};
}

// ---- shapes (shapes/triangle_mesh.inl, shapes/sphere.inl)
inline Vector3 mesh_p(const OMesh &m, int i) { return {m.P[3 * i], m.P[3 * i + 1], m.P[3 * i + 2]}; }
inline Vector3 mesh_n(const OMesh &m, int i) { return {m.N[3 * i], m.N[3 * i + 1], m.N[3 * i + 2]}; }
inline Vector2 mesh_uv(const OMesh &m, int i) { return {m.UV[2 * i], m.UV[2 * i + 1]}; }

Real surface_area(const OScene &s, int shape_id) {
    const LjShape &sh = s.shapes[shape_id];
    if (sh.kind == LJ_SHAPE_SPHERE) return 4 * c_PI * sh.radius * sh.radius;  // sphere.inl:206-208
    return s.meshes[shape_id].total_area;                                      // triangle_mesh.inl:40-42
}

PointAndNormal sample_point_on_shape(const OScene &s, int shape_id, const Vector3 &ref_point, const Vector2 &uv, Real w) {
    const LjShape &sh = s.shapes[shape_id];
    if (sh.kind == LJ_SHAPE_TRIMESH) {  // triangle_mesh.inl:24-38
        const OMesh &m = s.meshes[shape_id];
        int tri = sample_1d(m.tri_sampler, w);
        Vector3 v0 = mesh_p(m, m.I[3 * tri]), v1 = mesh_p(m, m.I[3 * tri + 1]), v2 = mesh_p(m, m.I[3 * tri + 2]);
        Vector3 e1 = v1 - v0, e2 = v2 - v0;
        Real a = std::sqrt(clampr(uv.x, Real(0), Real(1)));
        Real b1 = 1 - a, b2 = a * uv.y;
        return {v0 + (e1 * b1) + (e2 * b2), normalize(cross(e1, e2))};
    }
    // sphere.inl:156-204
    Vector3 center{sh.position[0], sh.position[1], sh.position[2]}; Real r = sh.radius;
    if (distance_squared(ref_point, center) < r * r) {
        Real z = 1 - 2 * uv.x;
        Real r_ = std::sqrt(std::fmax(Real(0), 1 - z * z));
        Real phi = 2 * c_PI * uv.y;
        Vector3 offset{r_ * std::cos(phi), r_ * std::sin(phi), z};
        return {center + r * offset, offset};
    }
    Vector3 dir_to_center = normalize(center - ref_point);
    Frame frame = make_frame(dir_to_center);
    Real sin_elevation_max_sq = r * r / distance_squared(ref_point, center);
    Real cos_elevation_max = std::sqrt(std::max(Real(0), 1 - sin_elevation_max_sq));
    Real cos_elevation = (1 - uv.x) + uv.x * cos_elevation_max;
    Real sin_elevation = std::sqrt(std::max(Real(0), 1 - cos_elevation * cos_elevation));
    Real azimuth = uv.y * 2 * c_PI;
    Real dc = distance(ref_point, center);
    Real ds = dc * cos_elevation - std::sqrt(std::max(Real(0), r * r - dc * dc * sin_elevation * sin_elevation));
    Real cos_alpha = (dc * dc + r * r - ds * ds) / (2 * dc * r);
    Real sin_alpha = std::sqrt(std::max(Real(0), 1 - cos_alpha * cos_alpha));
    Vector3 n_on_sphere = -to_world(frame, Vector3{sin_alpha * std::cos(azimuth), sin_alpha * std::sin(azimuth), cos_alpha});
    return {r * n_on_sphere + center, n_on_sphere};
}

Real pdf_point_on_shape(const OScene &s, int shape_id, const PointAndNormal &pn, const Vector3 &ref_point) {
    const LjShape &sh = s.shapes[shape_id];
    if (sh.kind == LJ_SHAPE_TRIMESH) return 1 / s.meshes[shape_id].total_area;  // triangle_mesh.inl:44-46
    // sphere.inl:210-230
    Vector3 center{sh.position[0], sh.position[1], sh.position[2]}; Real r = sh.radius;
    if (distance_squared(ref_point, center) < r * r) return 1 / surface_area(s, shape_id);
    Real sin_elevation_max_sq = r * r / distance_squared(ref_point, center);
    Real cos_elevation_max = std::sqrt(std::max(Real(0), 1 - sin_elevation_max_sq));
    Real pdf_solid_angle = 1 / (2 * c_PI * (1 - cos_elevation_max));
    Vector3 dir = normalize(pn.position - ref_point);
    return pdf_solid_angle * std::fabs(dot(pn.normal, dir)) / distance_squared(ref_point, pn.position);
}

struct ShadingInfo { Vector2 uv; Frame shading_frame; Real mean_curvature, inv_uv_size; };
ShadingInfo compute_shading_info(const OScene &s, const PathVertex &vertex) {
    const LjShape &sh = s.shapes[vertex.shape_id];
    if (sh.kind == LJ_SHAPE_SPHERE) {  // sphere.inl:235-260 (st consumed as radians, as written)
        Real r = sh.radius;
        Vector3 dpdu{-r * std::sin(vertex.st.x) * std::sin(vertex.st.y), r * std::cos(vertex.st.x) * std::sin(vertex.st.y), Real(0)};
        Vector3 dpdv{r * std::cos(vertex.st.x) * std::cos(vertex.st.y), r * std::sin(vertex.st.x) * std::cos(vertex.st.y), -r * std::sin(vertex.st.y)};
        Vector3 tangent = normalize(dpdu - vertex.geometry_normal * dot(vertex.geometry_normal, dpdu));
        Frame f{tangent, normalize(cross(vertex.geometry_normal, tangent)), vertex.geometry_normal};
        return {vertex.st, f, 1 / r, (length(dpdu) + length(dpdv)) / 2};
    }
    // triangle_mesh.inl:65-157
    const OMesh &m = s.meshes[vertex.shape_id];
    int i0 = m.I[3 * vertex.primitive_id], i1 = m.I[3 * vertex.primitive_id + 1], i2 = m.I[3 * vertex.primitive_id + 2];
    Vector2 uvs[3];
    if (m.has_uv) { uvs[0] = mesh_uv(m, i0); uvs[1] = mesh_uv(m, i1); uvs[2] = mesh_uv(m, i2); }
    else { uvs[0] = {0, 0}; uvs[1] = {1, 0}; uvs[2] = {1, 1}; }
    Real b0 = 1 - vertex.st.x - vertex.st.y;
    Vector2 uv{b0 * uvs[0].x + vertex.st.x * uvs[1].x + vertex.st.y * uvs[2].x, b0 * uvs[0].y + vertex.st.x * uvs[1].y + vertex.st.y * uvs[2].y};
    Vector3 p0 = mesh_p(m, i0), p1 = mesh_p(m, i1), p2 = mesh_p(m, i2);
    Vector2 duvds{uvs[2].x - uvs[0].x, uvs[2].y - uvs[0].y}, duvdt{uvs[2].x - uvs[1].x, uvs[2].y - uvs[1].y};
    Real det = duvds.x * duvdt.y - duvdt.x * duvds.y;
    Real dsdu = duvdt.y / det, dtdu = -duvds.y / det, dsdv = duvdt.x / det, dtdv = -duvds.x / det;
    Vector3 dpdu, dpdv;
    if (std::fabs(det) > 1e-8f) {
        Vector3 dpds = p2 - p0, dpdt = p2 - p1;
        dpdu = dpds * dsdu + dpdt * dtdu; dpdv = dpds * dsdv + dpdt * dtdv;
    } else coordinate_system(vertex.geometry_normal, dpdu, dpdv);
    Vector3 shading_normal = vertex.geometry_normal; Real mean_curvature = 0; Vector3 tangent, bitangent;
    if (m.has_n) {
        Vector3 n0 = mesh_n(m, i0), n1 = mesh_n(m, i1), n2 = mesh_n(m, i2);
        shading_normal = normalize(b0 * n0 + vertex.st.x * n1 + vertex.st.y * n2);
        tangent = normalize(dpdu - shading_normal * dot(shading_normal, dpdu));
        Vector3 dnds = n2 - n0, dndt = n2 - n1;
        Vector3 dndu = dnds * dsdu + dndt * dtdu, dndv = dnds * dsdv + dndt * dtdv;
        bitangent = normalize(cross(shading_normal, tangent));
        mean_curvature = (dot(dndu, tangent) + dot(dndv, bitangent)) / Real(2);
    } else {
        tangent = normalize(dpdu - shading_normal * dot(shading_normal, dpdu));
        bitangent = normalize(cross(shading_normal, tangent));
    }
    return {uv, Frame{tangent, bitangent, shading_normal}, mean_curvature, std::max(length(dpdu), length(dpdv))};
}

// intersection.cpp:38-62 given the hit record
PathVertex make_vertex(const OScene &s, const Ray &ray, const RayDifferential &rd, int shape_id, int prim_id, float t, float u, float v, const Vector3 &Ng) {
    PathVertex vx;
    vx.position = ray.org + ray.dir * Real(t);
    vx.geometry_normal = normalize(Ng);
    vx.shape_id = shape_id; vx.primitive_id = prim_id; vx.material_id = s.shapes[shape_id].material_id;
    vx.st = {Real(u), Real(v)};
    ShadingInfo si = compute_shading_info(s, vx);
    vx.shading_frame = si.shading_frame; vx.uv = si.uv; vx.mean_curvature = si.mean_curvature;
    vx.ray_radius = rd_transfer(rd, distance(ray.org, vx.position));
    vx.uv_screen_size = vx.ray_radius / si.inv_uv_size;
    if (dot(vx.geometry_normal, vx.shading_frame.n) < 0) vx.geometry_normal = -vx.geometry_normal;
    return vx;
}

// ---- lights (lights/diffuse_area_light.inl, lights/envmap.inl)
Real light_power(const OScene &s, int light_id) {
    const LjLight &l = s.lights[light_id];
    if (l.kind == LJ_LIGHT_AREA) return luminance({l.intensity[0], l.intensity[1], l.intensity[2]}) * surface_area(s, l.shape_id) * c_PI;  // :1-3
    const TableDist2D &d = s.env_dist[light_id];
    return c_PI * s.bounds_radius * s.bounds_radius * d.total_values / (d.width * d.height);  // envmap.inl:1-5
}
PointAndNormal sample_point_on_light(const OScene &s, int light_id, const Vector3 &ref, const Vector2 &uv, Real w) {
    const LjLight &l = s.lights[light_id];
    if (l.kind == LJ_LIGHT_AREA) return sample_point_on_shape(s, l.shape_id, ref, uv, w);
    Vector2 xy = sample_2d(s.env_dist[light_id], uv);  // envmap.inl:7-20
    Real azimuth = xy.x * (2 * c_PI), elevation = xy.y * c_PI;
    Vector3 local_dir{std::sin(azimuth) * std::sin(elevation), std::cos(elevation), -std::cos(azimuth) * std::sin(elevation)};
    Matrix4x4 tw; memcpy(tw.m, l.to_world, sizeof tw.m);
    Vector3 world_dir = xform_vector(tw, local_dir);
    return {Vector3{0, 0, 0}, -world_dir};
}
static Vector2 envmap_dir_to_uv(const Vector3 &local_dir) {
    Vector2 uv{std::atan2(local_dir.x, -local_dir.z) * c_INVTWOPI, std::acos(clampr(local_dir.y, Real(-1), Real(1))) * c_INVPI};
    if (uv.x < 0) uv.x += 1;
    return uv;
}
Real pdf_point_on_light(const OScene &s, int light_id, const PointAndNormal &pn, const Vector3 &ref) {
    const LjLight &l = s.lights[light_id];
    if (l.kind == LJ_LIGHT_AREA) return pdf_point_on_shape(s, l.shape_id, pn, ref);
    Matrix4x4 tl; memcpy(tl.m, l.to_local, sizeof tl.m);  // envmap.inl:22-42
    Vector3 local_dir = xform_vector(tl, -pn.normal);
    Vector2 uv = envmap_dir_to_uv(local_dir);
    Real cos_elevation = local_dir.y;
    Real sin_elevation = std::sqrt(clampr(1 - cos_elevation * cos_elevation, Real(0), Real(1)));
    if (sin_elevation <= 0) return 0;
    return pdf_2d(s.env_dist[light_id], uv) / (2 * c_PI * c_PI * sin_elevation);
}
Spectrum light_emission(const OScene &s, int light_id, const Vector3 &view_dir, Real view_footprint, const PointAndNormal &pn) {
    const LjLight &l = s.lights[light_id];
    if (l.kind == LJ_LIGHT_AREA) {  // diffuse_area_light.inl:15-20
        if (dot(pn.normal, view_dir) <= 0) return {0, 0, 0};
        return {l.intensity[0], l.intensity[1], l.intensity[2]};
    }
    Matrix4x4 tl; memcpy(tl.m, l.to_local, sizeof tl.m);  // envmap.inl:44-73
    Vector3 w = xform_vector(tl, -view_dir);
    Vector2 uv = envmap_dir_to_uv(w);
    Real dudwx = -w.z / (w.x * w.x + w.z * w.z), dudwz = w.x / (w.x * w.x + w.z * w.z);
    Real dvdwy = -1 / std::sqrt(std::max(1 - w.y * w.y, Real(0)));
    (void)view_footprint;  // the reference computes `footprint` without view_footprint (envmap.inl:66-70), as written
    Real footprint = std::min(std::sqrt(dudwx * dudwx + dudwz * dudwz), dvdwy);
    return eval_texture(&s, l.values, true, uv, footprint) * l.scale;
}

// ---- materials (material.cpp:4-11, materials/*.inl, microfacet.h)
inline Vector3 sample_cos_hemisphere(const Vector2 &rnd) {  // material.cpp:4-11
    Real phi = c_TWOPI * rnd.x;
    Real tmp = std::sqrt(clampr(1 - rnd.y, Real(0), Real(1)));
    return {std::cos(phi) * tmp, std::sin(phi) * tmp, std::sqrt(clampr(rnd.y, Real(0), Real(1)))};
}
inline Real fresnel_dielectric2(Real n_dot_i, Real n_dot_t, Real eta) {  // microfacet.h:34-40
    Real rs = (n_dot_i - eta * n_dot_t) / (n_dot_i + eta * n_dot_t);
    Real rp = (eta * n_dot_i - n_dot_t) / (eta * n_dot_i + n_dot_t);
    return (rs * rs + rp * rp) / 2;
}
inline Real fresnel_dielectric(Real n_dot_i, Real eta) {  // microfacet.h:47-56
    Real n_dot_t_sq = 1 - (1 - n_dot_i * n_dot_i) / (eta * eta);
    if (n_dot_t_sq < 0) return 1;
    return fresnel_dielectric2(std::fabs(n_dot_i), std::sqrt(n_dot_t_sq), eta);
}
inline Real GTR2(Real n_dot_h, Real roughness) {  // microfacet.h:58-63
    Real alpha = roughness * roughness, a2 = alpha * alpha;
    Real t = 1 + (a2 - 1) * n_dot_h * n_dot_h;
    return a2 / (c_PI * t * t);
}
inline Real smith_masking_gtr2(const Vector3 &v_local, Real roughness) {  // microfacet.h:75-81
    Real alpha = roughness * roughness, a2 = alpha * alpha;
    Vector3 v2 = v_local * v_local;
    Real Lambda = (-1 + std::sqrt(1 + (v2.x * a2 + v2.y * a2) / v2.z)) / 2;
    return 1 / (1 + Lambda);
}
Vector3 sample_visible_normals(const Vector3 &local_dir_in, Real alpha, const Vector2 &rnd) {  // microfacet.h:85-114
    if (local_dir_in.z < 0) return -sample_visible_normals(-local_dir_in, alpha, rnd);
    Vector3 hemi_dir_in = normalize(Vector3{alpha * local_dir_in.x, alpha * local_dir_in.y, local_dir_in.z});
    Real r = std::sqrt(rnd.x), phi = 2 * c_PI * rnd.y;
    Real t1 = r * std::cos(phi), t2 = r * std::sin(phi);
    Real s = (1 + hemi_dir_in.z) / 2;
    t2 = (1 - s) * std::sqrt(1 - t1 * t1) + s * t2;
    Vector3 disk_N{t1, t2, std::sqrt(std::max(Real(0), 1 - t1 * t1 - t2 * t2))};
    Frame hemi_frame = make_frame(hemi_dir_in);
    Vector3 hemi_N = to_world(hemi_frame, disk_N);
    return normalize(Vector3{alpha * hemi_N.x, alpha * hemi_N.y, std::max(Real(0), hemi_N.z)});
}

inline Spectrum tex3(const OScene *s, const LjMaterial &m, int slot, const PathVertex &v) { return eval_texture(s, m.tex[slot], true, v.uv, v.uv_screen_size); }
struct BSDFSampleRecord { Vector3 dir_out; Real eta, roughness; };
inline Real tex1(const OScene *s, const LjMaterial &m, int slot, const PathVertex &v) { return eval_texture(s, m.tex[slot], false, v.uv, v.uv_screen_size).x; }

// ---- Disney BSDF family helpers (materials/disney_metal.inl:3-50, disney_clearcoat.inl:3-16)
inline Real smithG_GGX_aniso(Real NdotW, Real WdotX, Real WdotY, Real ax, Real ay) {  // disney_metal.inl:3-8
    Real lambda = Real(0.5) * (std::sqrt(Real(1) + (std::pow(WdotX * ax, 2) + std::pow(WdotY * ay, 2)) / std::pow(NdotW, 2)) - Real(1));
    return Real(1) / (Real(1) + lambda);
}
inline Real GTR2_aniso(Real ax, Real ay, const Frame &frame, const Vector3 &h) {  // disney_metal.inl:10-19
    Real ax2 = ax * ax, ay2 = ay * ay;
    Real hlx2 = std::pow(dot(frame.x, h), 2), hly2 = std::pow(dot(frame.y, h), 2), hlz2 = std::pow(dot(frame.n, h), 2);
    return Real(1) / (c_PI * ax * ay * std::pow(hlx2 / ax2 + hly2 / ay2 + hlz2, 2));
}
Vector3 sample_visible_normals_aniso(const Vector3 &local_dir_in, Real ax, Real ay, const Vector2 &rnd) {  // disney_metal.inl:21-50
    if (local_dir_in.z < 0) return -sample_visible_normals_aniso(-local_dir_in, ax, ay, rnd);
    Vector3 hemi_dir_in = normalize(Vector3{ax * local_dir_in.x, ay * local_dir_in.y, local_dir_in.z});
    Real r = std::sqrt(rnd.x), phi = 2 * c_PI * rnd.y;
    Real t1 = r * std::cos(phi), t2 = r * std::sin(phi);
    Real s = (1 + hemi_dir_in.z) / 2;
    t2 = (1 - s) * std::sqrt(1 - t1 * t1) + s * t2;
    Vector3 disk_N{t1, t2, std::sqrt(std::max(Real(0), 1 - t1 * t1 - t2 * t2))};
    Vector3 hemi_N = to_world(make_frame(hemi_dir_in), disk_N);
    return normalize(Vector3{ax * hemi_N.x, ay * hemi_N.y, std::max(Real(0), hemi_N.z)});
}
inline Real clearcoat_schlick_fresnel(const Vector3 &half_vector, const Vector3 &dir_out) {  // disney_clearcoat.inl:3-8
    Real eta = Real(1.5);
    Real R_0 = std::pow(eta - Real(1), 2) / std::pow(eta + Real(1), 2);
    return R_0 + (Real(1) - R_0) * std::pow(Real(1) - std::fabs(dot(half_vector, dir_out)), 5);
}
inline Real compute_Dc(Real clearcoat_gloss, Real hlz2) {  // disney_clearcoat.inl:10-16
    Real a = (Real(1) - clearcoat_gloss) * Real(0.1) + clearcoat_gloss * Real(0.001);
    Real a2 = a * a;
    return (a2 - Real(1)) / (c_PI * std::log(a2) * (Real(1) + (a2 - Real(1)) * hlz2));
}
inline void aniso_alphas(Real roughness, Real anisotropic, Real &ax, Real &ay) {  // disney_metal.inl:75-78 (and glass / bsdf copies)
    Real aspect = std::sqrt(Real(1) - Real(0.9) * anisotropic);
    Real a_min = Real(0.0001);
    ax = std::fmax(a_min, roughness * roughness / aspect);
    ay = std::fmax(a_min, roughness * roughness * aspect);
}
inline Frame frame_to_dir_in(const PathVertex &vertex, const Vector3 &dir_in) {  // "flip the shading frame" idiom (lambertian.inl:10-13)
    Frame frame = vertex.shading_frame;
    if (dot(frame.n, dir_in) < 0) frame = -frame;
    return frame;
}
inline Frame frame_two_sided(const PathVertex &vertex, const Vector3 &dir_in) {  // roughdielectric.inl:6-9 / disney_glass.inl:6-9
    Frame frame = vertex.shading_frame;
    if (dot(frame.n, dir_in) * dot(vertex.geometry_normal, dir_in) < 0) frame = -frame;
    return frame;
}
// the five lobes of disney_bsdf.inl, which are also (verbatim) the bodies of the five standalone Disney materials
Spectrum disney_diffuse_lobe(const Spectrum &base_color, Real roughness, Real subsurface, const Frame &frame, const Vector3 &dir_in, const Vector3 &dir_out) {
    Vector3 half_vector = normalize(dir_in + dir_out);  // disney_diffuse.inl:19-39 / disney_bsdf.inl:33-52
    Real h_dot_out = dot(half_vector, dir_out), n_dot_in = dot(frame.n, dir_in), n_dot_out = dot(frame.n, dir_out);
    Real FD90 = Real(0.5) + Real(2) * roughness * h_dot_out * h_dot_out;
    Real FD_in = Real(1) + (FD90 - Real(1)) * (Real(1) - std::pow(n_dot_in, 5));   // pow(n.w, 5), not pow(1 - n.w, 5): as written
    Real FD_out = Real(1) + (FD90 - Real(1)) * (Real(1) - std::pow(n_dot_out, 5));
    Spectrum f_d = base_color * FD_in * FD_out * std::fabs(n_dot_out) / c_PI;
    Real FSS90 = roughness * h_dot_out * h_dot_out;
    Real FSS_in = Real(1) + (FSS90 - Real(1)) * (Real(1) - std::pow(n_dot_in, 5));
    Real FSS_out = Real(1) + (FSS90 - Real(1)) * (Real(1) - std::pow(n_dot_out, 5));
    Spectrum f_ss = Real(1.25) * base_color * (FSS_in * FSS_out * (Real(1) / (std::fabs(n_dot_in) + std::fabs(n_dot_out)) - Real(0.5)) + Real(0.5)) * std::fabs(n_dot_out) / c_PI;
    return (Real(1) - subsurface) * f_d + subsurface * f_ss;
}
Spectrum disney_glass_lobe(const Spectrum &base_color, Real roughness_raw, Real anisotropic, Real bsdf_eta, const PathVertex &vertex, const Vector3 &dir_in, const Vector3 &dir_out, Real *pdf_out) {
    bool reflect = dot(vertex.geometry_normal, dir_in) * dot(vertex.geometry_normal, dir_out) > 0;  // disney_glass.inl:3-83,85-135
    Frame frame = frame_two_sided(vertex, dir_in);
    Real eta = dot(vertex.geometry_normal, dir_in) > 0 ? bsdf_eta : 1 / bsdf_eta;
    Vector3 half_vector = reflect ? normalize(dir_in + dir_out) : normalize(dir_in + dir_out * eta);
    if (dot(half_vector, frame.n) < 0) half_vector = -half_vector;
    Real roughness = clampr(roughness_raw, Real(0.01), Real(1));
    Real ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
    Real h_dot_in = dot(half_vector, dir_in);
    Real F = fresnel_dielectric(h_dot_in, eta);
    Real D = GTR2_aniso(ax, ay, frame, half_vector);
    Real G = smithG_GGX_aniso(dot(dir_in, frame.n), dot(dir_in, frame.x), dot(dir_in, frame.y), ax, ay);  // G_in only (disney_glass.inl:59)
    if (reflect) {
        if (pdf_out) *pdf_out = (F * D * G) / (4 * std::fabs(dot(frame.n, dir_in)));
        return base_color * (F * D * G) / (4 * std::fabs(dot(frame.n, dir_in)));
    }
    Real h_dot_out = dot(half_vector, dir_out);
    if (pdf_out) {
        Real sqrt_denom = h_dot_in + eta * h_dot_out;
        Real dh_dout = eta * eta * h_dot_out / (sqrt_denom * sqrt_denom);
        *pdf_out = (1 - F) * D * G * std::fabs(dh_dout * h_dot_in / dot(frame.n, dir_in));
    }
    Spectrum sq{std::sqrt(std::max(base_color.x, Real(0))), std::sqrt(std::max(base_color.y, Real(0))), std::sqrt(std::max(base_color.z, Real(0)))};  // spectrum.h:22-26
    return sq * (Real(1) - F) * D * G * std::fabs(h_dot_out * h_dot_in) / (std::fabs(dot(frame.n, dir_in)) * std::pow(h_dot_in + eta * h_dot_out, 2));
}
// shared tail of the glass samplers (disney_glass.inl:167-205, disney_bsdf.inl:488-535, roughdielectric.inl:150-176)
bool sample_dielectric_tail(const Vector3 &dir_in, Vector3 half_vector, const Frame &frame, Real eta, Real roughness, Real rnd, BSDFSampleRecord &rec) {
    if (dot(half_vector, frame.n) < 0) half_vector = -half_vector;
    Real h_dot_in = dot(half_vector, dir_in);
    Real F = fresnel_dielectric(h_dot_in, eta);
    if (rnd <= F) {
        rec = {normalize(-dir_in + 2 * dot(dir_in, half_vector) * half_vector), Real(0), roughness};
        return true;
    }
    Real h_dot_out_sq = 1 - (1 - h_dot_in * h_dot_in) / (eta * eta);
    if (h_dot_out_sq <= 0) return false;
    if (h_dot_in < 0) half_vector = -half_vector;
    Real h_dot_out = std::sqrt(h_dot_out_sq);
    rec = {-dir_in / eta + (std::fabs(h_dot_in) / eta - h_dot_out) * half_vector, eta, roughness};
    return true;
}
inline Spectrum color_tint(const Spectrum &base_color) {  // disney_sheen.inl:24-25
    if (luminance(base_color) <= 0) return {1, 1, 1};
    return base_color / luminance(base_color);
}
inline Vector3 sample_clearcoat_half(Real clearcoat_gloss, const Vector2 &rnd) {  // disney_clearcoat.inl:85-97
    Real a = (Real(1) - clearcoat_gloss) * Real(0.1) + clearcoat_gloss * Real(0.001);
    Real a2 = a * a;
    Real cos_h_elevation = std::sqrt((Real(1) - std::pow(a2, Real(1) - rnd.x)) / (Real(1) - a2));
    Real h_elevation = std::acos(cos_h_elevation), h_azimuth = Real(2) * c_PI * rnd.y;
    return normalize(Vector3{std::sin(h_elevation) * std::cos(h_azimuth), std::sin(h_elevation) * std::sin(h_azimuth), std::cos(h_elevation)});
}

// returns false for material kinds the oracle does not restate yet
bool bsdf_eval(const OScene *s, const LjMaterial &m, const Vector3 &dir_in, const Vector3 &dir_out, const PathVertex &vertex, Spectrum &out, bool to_view = false) {
    out = {0, 0, 0};
    if (m.kind == LJ_MAT_LAMBERTIAN) {  // lambertian.inl:1-17
        if (dot(vertex.geometry_normal, dir_in) < 0 || dot(vertex.geometry_normal, dir_out) < 0) return true;
        Frame frame = vertex.shading_frame;
        if (dot(frame.n, dir_in) < 0) frame = -frame;
        out = std::fmax(dot(frame.n, dir_out), Real(0)) * tex3(s, m, 0, vertex) / c_PI;
        return true;
    }
    if (m.kind == LJ_MAT_ROUGHPLASTIC) {  // roughplastic.inl:3-63
        if (dot(vertex.geometry_normal, dir_in) < 0 || dot(vertex.geometry_normal, dir_out) < 0) return true;
        Frame frame = vertex.shading_frame;
        if (dot(frame.n, dir_in) < 0) frame = -frame;
        Vector3 half_vector = normalize(dir_in + dir_out);
        Real n_dot_h = dot(frame.n, half_vector), n_dot_in = dot(frame.n, dir_in), n_dot_out = dot(frame.n, dir_out);
        if (n_dot_out <= 0 || n_dot_h <= 0) return true;
        Spectrum Kd = tex3(s, m, 0, vertex), Ks = tex3(s, m, 1, vertex);
        Real roughness = clampr(tex1(s, m, 2, vertex), Real(0.01), Real(1));
        Real F_o = fresnel_dielectric(dot(half_vector, dir_out), m.eta);
        Real D = GTR2(n_dot_h, roughness);
        Real G = smith_masking_gtr2(to_local(frame, dir_in), roughness) * smith_masking_gtr2(to_local(frame, dir_out), roughness);
        Spectrum spec_contrib = Ks * (G * F_o * D) / (4 * n_dot_in * n_dot_out);
        Real F_i = fresnel_dielectric(dot(half_vector, dir_in), m.eta);
        Spectrum diffuse_contrib = Kd * (Real(1) - F_o) * (Real(1) - F_i) / c_PI;
        out = (spec_contrib + diffuse_contrib) * n_dot_out;
        return true;
    }

    const bool above = !(dot(vertex.geometry_normal, dir_in) < 0 || dot(vertex.geometry_normal, dir_out) < 0);
    if (m.kind == LJ_MAT_ROUGHDIELECTRIC) {  // roughdielectric.inl:3-47
        bool reflect = dot(vertex.geometry_normal, dir_in) * dot(vertex.geometry_normal, dir_out) > 0;
        Frame frame = frame_two_sided(vertex, dir_in);
        Real eta = dot(vertex.geometry_normal, dir_in) > 0 ? m.eta : 1 / m.eta;
        Spectrum Ks = tex3(s, m, 0, vertex), Kt = tex3(s, m, 1, vertex);
        Real roughness = tex1(s, m, 2, vertex);
        Vector3 half_vector = reflect ? normalize(dir_in + dir_out) : normalize(dir_in + dir_out * eta);
        if (dot(half_vector, frame.n) < 0) half_vector = -half_vector;
        roughness = clampr(roughness, Real(0.01), Real(1));
        Real h_dot_in = dot(half_vector, dir_in);
        Real F = fresnel_dielectric(h_dot_in, eta);
        Real D = GTR2(dot(frame.n, half_vector), roughness);
        Real G = smith_masking_gtr2(to_local(frame, dir_in), roughness) * smith_masking_gtr2(to_local(frame, dir_out), roughness);
        if (reflect) out = Ks * (F * D * G) / (4 * std::fabs(dot(frame.n, dir_in)));
        else {
            Real eta_factor = to_view ? 1 : (1 / (eta * eta));
            Real h_dot_out = dot(half_vector, dir_out);
            Real sqrt_denom = h_dot_in + eta * h_dot_out;
            out = Kt * (eta_factor * (1 - F) * D * G * eta * eta * std::fabs(h_dot_out * h_dot_in)) / (std::fabs(dot(frame.n, dir_in)) * sqrt_denom * sqrt_denom);
        }
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYDIFFUSE) {  // disney_diffuse.inl:1-42
        if (!above) return true;
        out = disney_diffuse_lobe(tex3(s, m, 0, vertex), tex1(s, m, 1, vertex), tex1(s, m, 2, vertex), frame_to_dir_in(vertex, dir_in), dir_in, dir_out);
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYMETAL) {  // disney_metal.inl:52-93
        if (!above) return true;
        Frame frame = frame_to_dir_in(vertex, dir_in);
        Spectrum base_color = tex3(s, m, 0, vertex);
        Real roughness = clampr(tex1(s, m, 1, vertex), Real(0.01), Real(1)), anisotropic = tex1(s, m, 2, vertex);
        Vector3 half_vector = normalize(dir_in + dir_out);
        Real h_dot_out = dot(half_vector, dir_out);
        Spectrum Fm = base_color + (Vector3{1, 1, 1} - base_color) * std::pow(Real(1) - std::fabs(h_dot_out), 5);
        Real ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
        Real Dm = GTR2_aniso(ax, ay, frame, half_vector);
        Real Gin = smithG_GGX_aniso(dot(dir_in, frame.n), dot(dir_in, frame.x), dot(dir_in, frame.y), ax, ay);
        Real Gout = smithG_GGX_aniso(dot(dir_out, frame.n), dot(dir_out, frame.x), dot(dir_out, frame.y), ax, ay);
        out = Fm * Dm * Gin * Gout / (Real(4) * std::fabs(dot(dir_in, frame.n)));
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYGLASS) {  // disney_glass.inl:3-83
        out = disney_glass_lobe(tex3(s, m, 0, vertex), tex1(s, m, 1, vertex), tex1(s, m, 2, vertex), m.eta, vertex, dir_in, dir_out, nullptr);
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYCLEARCOAT) {  // disney_clearcoat.inl:18-45
        if (!above) return true;
        Frame frame = frame_to_dir_in(vertex, dir_in);
        Vector3 half_vector = normalize(dir_in + dir_out);
        Real n_dot_h = dot(frame.n, half_vector), n_dot_in = dot(frame.n, dir_in);
        if (n_dot_h <= 0) return true;
        Real F = clearcoat_schlick_fresnel(half_vector, dir_out);
        Real D = compute_Dc(tex1(s, m, 0, vertex), std::pow(dot(frame.n, half_vector), 2));
        Real G = smith_masking_gtr2(to_local(frame, dir_in), Real(0.5)) * smith_masking_gtr2(to_local(frame, dir_out), Real(0.5));
        Real v = F * D * G / (Real(4) * std::fabs(n_dot_in));
        out = {v, v, v};
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYSHEEN) {  // disney_sheen.inl:3-31
        if (!above) return true;
        Frame frame = frame_to_dir_in(vertex, dir_in);
        Spectrum base_color = tex3(s, m, 0, vertex); Real sheen_tint = tex1(s, m, 1, vertex);
        Vector3 half_vector = normalize(dir_in + dir_out);
        Real n_dot_out = dot(frame.n, dir_out);
        Spectrum C_sheen = Vector3{1, 1, 1} * (Real(1) - sheen_tint) + sheen_tint * color_tint(base_color);
        out = C_sheen * std::pow(Real(1) - std::fabs(dot(half_vector, dir_out)), 5) * std::fabs(n_dot_out);
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYBSDF) {  // disney_bsdf.inl:3-218
        Spectrum base_color = tex3(s, m, 0, vertex);
        Real specular_transmission = tex1(s, m, 1, vertex), metallic = tex1(s, m, 2, vertex), subsurface = tex1(s, m, 3, vertex), specular = tex1(s, m, 4, vertex);
        Real roughness_raw = tex1(s, m, 5, vertex), specular_tint = tex1(s, m, 6, vertex), anisotropic = tex1(s, m, 7, vertex), sheen = tex1(s, m, 8, vertex);
        Real sheen_tint = tex1(s, m, 9, vertex), clearcoat = tex1(s, m, 10, vertex);
        Spectrum f_diffuse{0, 0, 0}, f_metal{0, 0, 0}, f_glass{0, 0, 0}, f_clearcoat{0, 0, 0}, f_sheen{0, 0, 0};
        if (dot(vertex.geometry_normal, dir_in) >= 0 && dot(vertex.geometry_normal, dir_out) >= 0) {
            Frame frame = frame_to_dir_in(vertex, dir_in);
            f_diffuse = disney_diffuse_lobe(base_color, roughness_raw, subsurface, frame, dir_in, dir_out);
            {   // metal lobe with the achromatic specular blend (disney_bsdf.inl:56-88); Fresnel without fabs, as written (:77)
                Vector3 half_vector = normalize(dir_in + dir_out);
                Real roughness = clampr(roughness_raw, Real(0.01), Real(1));
                Real h_dot_out = dot(half_vector, dir_out);
                Spectrum C_tint = color_tint(base_color);
                Real eta = Real(1.5);
                Real R_0 = std::pow(eta - Real(1), 2) / std::pow(eta + Real(1), 2);
                Spectrum Ks = Vector3{1, 1, 1} * (Real(1) - specular_tint) + specular_tint * C_tint;
                Spectrum C0 = specular * R_0 * (Real(1) - metallic) * Ks + metallic * base_color;
                Spectrum Fm = C0 + (Vector3{1, 1, 1} - C0) * std::pow(Real(1) - h_dot_out, 5);
                Real ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
                Real Dm = GTR2_aniso(ax, ay, frame, half_vector);
                Real Gin = smithG_GGX_aniso(dot(dir_in, frame.n), dot(dir_in, frame.x), dot(dir_in, frame.y), ax, ay);
                Real Gout = smithG_GGX_aniso(dot(dir_out, frame.n), dot(dir_out, frame.x), dot(dir_out, frame.y), ax, ay);
                f_metal = Fm * Dm * Gin * Gout / (Real(4) * std::fabs(dot(dir_in, frame.n)));
            }
            {   // clearcoat (disney_bsdf.inl:91-110)
                Vector3 half_vector = normalize(dir_in + dir_out);
                Real n_dot_h = dot(frame.n, half_vector), n_dot_in = dot(frame.n, dir_in);
                if (n_dot_h > 0) {
                    Real F = clearcoat_schlick_fresnel(half_vector, dir_out);
                    Real D = compute_Dc(tex1(s, m, 11, vertex), std::pow(dot(frame.n, half_vector), 2));
                    Real G = smith_masking_gtr2(to_local(frame, dir_in), Real(0.5)) * smith_masking_gtr2(to_local(frame, dir_out), Real(0.5));
                    Real v = F * D * G / (Real(4) * std::fabs(n_dot_in));
                    f_clearcoat = {v, v, v};
                }
            }
            {   // sheen (disney_bsdf.inl:113-128)
                Vector3 half_vector = normalize(dir_in + dir_out);
                Real n_dot_out = dot(frame.n, dir_out);
                Spectrum C_sheen = Vector3{1, 1, 1} * (Real(1) - sheen_tint) + sheen_tint * color_tint(base_color);
                f_sheen = C_sheen * std::pow(Real(1) - std::fabs(dot(half_vector, dir_out)), 5) * std::fabs(n_dot_out);
            }
        }
        f_glass = disney_glass_lobe(base_color, roughness_raw, anisotropic, m.eta, vertex, dir_in, dir_out, nullptr);  // disney_bsdf.inl:131-175
        if (dot(vertex.geometry_normal, dir_in) < 0) { f_diffuse = f_metal = f_sheen = f_clearcoat = Spectrum{0, 0, 0}; }
        out = (Real(1) - specular_transmission) * (Real(1) - metallic) * f_diffuse
            + (Real(1) - metallic) * sheen * f_sheen
            + (Real(1) - specular_transmission * (Real(1) - metallic)) * f_metal
            + Real(0.25) * clearcoat * f_clearcoat
            + (Real(1) - metallic) * specular_transmission * f_glass;
        return true;
    }
    return false;
}
bool bsdf_pdf(const OScene *s, const LjMaterial &m, const Vector3 &dir_in, const Vector3 &dir_out, const PathVertex &vertex, Real &out) {
    out = 0;
    if (m.kind == LJ_MAT_LAMBERTIAN) {  // lambertian.inl:19-33
        if (dot(vertex.geometry_normal, dir_in) < 0 || dot(vertex.geometry_normal, dir_out) < 0) return true;
        Frame frame = vertex.shading_frame;
        if (dot(frame.n, dir_in) < 0) frame = -frame;
        out = std::fmax(dot(frame.n, dir_out), Real(0)) / c_PI;
        return true;
    }
    if (m.kind == LJ_MAT_ROUGHPLASTIC) {  // roughplastic.inl:65-108
        if (dot(vertex.geometry_normal, dir_in) < 0 || dot(vertex.geometry_normal, dir_out) < 0) return true;
        Frame frame = vertex.shading_frame;
        if (dot(frame.n, dir_in) < 0) frame = -frame;
        Vector3 half_vector = normalize(dir_in + dir_out);
        Real n_dot_in = dot(frame.n, dir_in), n_dot_out = dot(frame.n, dir_out), n_dot_h = dot(frame.n, half_vector);
        if (n_dot_out <= 0 || n_dot_h <= 0) return true;
        Spectrum S = tex3(s, m, 1, vertex), R = tex3(s, m, 0, vertex);
        Real lS = luminance(S), lR = luminance(R);
        if (lS + lR <= 0) return true;
        Real roughness = clampr(tex1(s, m, 2, vertex), Real(0.01), Real(1));
        Real spec_prob = lS / (lS + lR), diff_prob = 1 - spec_prob;
        Real G = smith_masking_gtr2(to_local(frame, dir_in), roughness), D = GTR2(n_dot_h, roughness);
        spec_prob *= (G * D) / (4 * n_dot_in);
        diff_prob *= n_dot_out / c_PI;
        out = spec_prob + diff_prob;
        return true;
    }

    const bool above = !(dot(vertex.geometry_normal, dir_in) < 0 || dot(vertex.geometry_normal, dir_out) < 0);
    if (m.kind == LJ_MAT_ROUGHDIELECTRIC) {  // roughdielectric.inl:49-88
        bool reflect = dot(vertex.geometry_normal, dir_in) * dot(vertex.geometry_normal, dir_out) > 0;
        Frame frame = frame_two_sided(vertex, dir_in);
        Real eta = dot(vertex.geometry_normal, dir_in) > 0 ? m.eta : 1 / m.eta;
        Vector3 half_vector = reflect ? normalize(dir_in + dir_out) : normalize(dir_in + dir_out * eta);
        if (dot(half_vector, frame.n) < 0) half_vector = -half_vector;
        Real roughness = clampr(tex1(s, m, 2, vertex), Real(0.01), Real(1));
        Real h_dot_in = dot(half_vector, dir_in);
        Real F = fresnel_dielectric(h_dot_in, eta);
        Real D = GTR2(dot(half_vector, frame.n), roughness);
        Real G_in = smith_masking_gtr2(to_local(frame, dir_in), roughness);
        if (reflect) out = (F * D * G_in) / (4 * std::fabs(dot(frame.n, dir_in)));
        else {
            Real h_dot_out = dot(half_vector, dir_out);
            Real sqrt_denom = h_dot_in + eta * h_dot_out;
            Real dh_dout = eta * eta * h_dot_out / (sqrt_denom * sqrt_denom);
            out = (1 - F) * D * G_in * std::fabs(dh_dout * h_dot_in / dot(frame.n, dir_in));
        }
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYDIFFUSE || m.kind == LJ_MAT_DISNEYSHEEN) {  // disney_diffuse.inl:44-58, disney_sheen.inl:33-46
        if (!above) return true;
        out = std::fmax(dot(frame_to_dir_in(vertex, dir_in).n, dir_out), Real(0)) / c_PI;
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYMETAL) {  // disney_metal.inl:95-126
        if (!above) return true;
        Frame frame = frame_to_dir_in(vertex, dir_in);
        Real roughness = clampr(tex1(s, m, 1, vertex), Real(0.01), Real(1)), anisotropic = tex1(s, m, 2, vertex);
        Vector3 half_vector = normalize(dir_in + dir_out);
        Real ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
        Real Dm = GTR2_aniso(ax, ay, frame, half_vector);
        Real Gin = smithG_GGX_aniso(dot(dir_in, frame.n), dot(dir_in, frame.x), dot(dir_in, frame.y), ax, ay);
        out = Dm * Gin / (Real(4) * std::fabs(dot(dir_in, frame.n)));
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYGLASS) {  // disney_glass.inl:85-135
        disney_glass_lobe(Spectrum{1, 1, 1}, tex1(s, m, 1, vertex), tex1(s, m, 2, vertex), m.eta, vertex, dir_in, dir_out, &out);
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYCLEARCOAT) {  // disney_clearcoat.inl:47-66
        if (!above) return true;
        Frame frame = frame_to_dir_in(vertex, dir_in);
        Vector3 half_vector = normalize(dir_in + dir_out);
        Real n_dot_h = dot(frame.n, half_vector);
        Real D = compute_Dc(tex1(s, m, 0, vertex), std::pow(dot(frame.n, half_vector), 2));
        out = D * std::fabs(n_dot_h) / (Real(4) * std::fabs(dot(half_vector, dir_out)));
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYBSDF) {  // disney_bsdf.inl:220-372
        bool reflect = dot(vertex.geometry_normal, dir_in) * dot(vertex.geometry_normal, dir_out) > 0;
        Real specular_transmission = tex1(s, m, 1, vertex), metallic = tex1(s, m, 2, vertex), anisotropic = tex1(s, m, 7, vertex), clearcoat = tex1(s, m, 10, vertex);
        Real diffuse_weight = (Real(1) - metallic) * (Real(1) - specular_transmission);
        Real metal_weight = (Real(1) - specular_transmission * (Real(1) - metallic));
        Real glass_weight = (Real(1) - metallic) * specular_transmission;
        Real clearcoat_weight = Real(0.25) * clearcoat;
        if (dot(vertex.geometry_normal, dir_in) < 0) {
            diffuse_weight = metal_weight = clearcoat_weight = 0;
            if (glass_weight > 0) glass_weight = 1; else return true;
        }
        Real weight_total = diffuse_weight + metal_weight + glass_weight + clearcoat_weight;
        diffuse_weight /= weight_total; metal_weight /= weight_total; glass_weight /= weight_total; clearcoat_weight /= weight_total;
        Frame frame = frame_to_dir_in(vertex, dir_in);
        Real diffuse_pdf = std::fmax(dot(frame.n, dir_out), Real(0)) / c_PI;
        Vector3 half_vector = normalize(dir_in + dir_out);
        Real roughness_raw = tex1(s, m, 5, vertex);
        Real roughness = clampr(roughness_raw, Real(0.01), Real(1));
        Real ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
        Real Dm = GTR2_aniso(ax, ay, frame, half_vector);
        Real Gin = smithG_GGX_aniso(dot(dir_in, frame.n), dot(dir_in, frame.x), dot(dir_in, frame.y), ax, ay);
        Real metal_pdf = Dm * Gin / (Real(4) * std::fabs(dot(dir_in, frame.n)));
        Real n_dot_h = dot(frame.n, half_vector);
        Real Dc = compute_Dc(tex1(s, m, 11, vertex), std::pow(dot(frame.n, half_vector), 2));
        Real clearcoat_pdf = Dc * std::fabs(n_dot_h) / (Real(4) * std::fabs(dot(half_vector, dir_out)));
        Real glass_pdf = 0;
        disney_glass_lobe(Spectrum{1, 1, 1}, roughness_raw, anisotropic, m.eta, vertex, dir_in, dir_out, &glass_pdf);
        if (reflect) out = diffuse_weight * diffuse_pdf + metal_weight * metal_pdf + clearcoat_weight * clearcoat_pdf + glass_weight * glass_pdf;
        else out = glass_weight * glass_pdf;
        return true;
    }
    return false;
}
// valid=false <=> std::nullopt
bool bsdf_sample(const OScene *s, const LjMaterial &m, const Vector3 &dir_in, const PathVertex &vertex, const Vector2 &rnd_uv, Real rnd_w,
                 bool &valid, BSDFSampleRecord &rec) {
    valid = false;
    if (m.kind == LJ_MAT_LAMBERTIAN) {  // lambertian.inl:35-50
        if (dot(vertex.geometry_normal, dir_in) < 0) return true;
        Frame frame = vertex.shading_frame;
        if (dot(frame.n, dir_in) < 0) frame = -frame;
        rec = {to_world(frame, sample_cos_hemisphere(rnd_uv)), Real(0), Real(1)};
        valid = true;
        return true;
    }
    if (m.kind == LJ_MAT_ROUGHPLASTIC) {  // roughplastic.inl:110-161
        if (dot(vertex.geometry_normal, dir_in) < 0) return true;
        Frame frame = vertex.shading_frame;
        if (dot(frame.n, dir_in) < 0) frame = -frame;
        Spectrum Ks = tex3(s, m, 1, vertex), Kd = tex3(s, m, 0, vertex);
        Real lS = luminance(Ks), lR = luminance(Kd);
        if (lS + lR <= 0) return true;
        Real spec_prob = lS / (lS + lR);
        if (rnd_w < spec_prob) {
            Vector3 local_dir_in = to_local(frame, dir_in);
            Real roughness = clampr(tex1(s, m, 2, vertex), Real(0.01), Real(1));
            Real alpha = roughness * roughness;
            Vector3 local_micro_normal = sample_visible_normals(local_dir_in, alpha, rnd_uv);
            Vector3 half_vector = to_world(frame, local_micro_normal);
            Vector3 reflected = normalize(-dir_in + 2 * dot(dir_in, half_vector) * half_vector);
            rec = {reflected, Real(0), roughness};
        } else rec = {to_world(frame, sample_cos_hemisphere(rnd_uv)), Real(0), Real(1)};
        valid = true;
        return true;
    }

    if (m.kind == LJ_MAT_ROUGHDIELECTRIC) {  // roughdielectric.inl:90-177
        Real eta = dot(vertex.geometry_normal, dir_in) > 0 ? m.eta : 1 / m.eta;
        Frame frame = frame_two_sided(vertex, dir_in);
        Real roughness = clampr(tex1(s, m, 2, vertex), Real(0.01), Real(1));
        Real alpha = roughness * roughness;
        Vector3 half_vector = to_world(frame, sample_visible_normals(to_local(frame, dir_in), alpha, rnd_uv));
        valid = sample_dielectric_tail(dir_in, half_vector, frame, eta, roughness, rnd_w, rec);
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYDIFFUSE || m.kind == LJ_MAT_DISNEYSHEEN) {  // disney_diffuse.inl:60-76, disney_sheen.inl:48-62
        if (dot(vertex.geometry_normal, dir_in) < 0) return true;
        rec = {to_world(frame_to_dir_in(vertex, dir_in), sample_cos_hemisphere(rnd_uv)), Real(0), Real(1)};
        valid = true;
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYMETAL) {  // disney_metal.inl:128-162
        if (dot(vertex.geometry_normal, dir_in) < 0) return true;
        Frame frame = frame_to_dir_in(vertex, dir_in);
        Real roughness = clampr(tex1(s, m, 1, vertex), Real(0.01), Real(1)), anisotropic = tex1(s, m, 2, vertex);
        Real ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
        Vector3 half_vector = to_world(frame, sample_visible_normals_aniso(to_local(frame, dir_in), ax, ay, rnd_uv));
        rec = {normalize(-dir_in + 2 * dot(dir_in, half_vector) * half_vector), Real(0), roughness};
        valid = true;
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYGLASS) {  // disney_glass.inl:137-205
        Frame frame = frame_two_sided(vertex, dir_in);
        Real eta = dot(vertex.geometry_normal, dir_in) > 0 ? m.eta : 1 / m.eta;
        Real anisotropic = tex1(s, m, 2, vertex), roughness = clampr(tex1(s, m, 1, vertex), Real(0.01), Real(1));
        Real ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
        Vector3 half_vector = to_world(frame, sample_visible_normals_aniso(to_local(frame, dir_in), ax, ay, rnd_uv));
        valid = sample_dielectric_tail(dir_in, half_vector, frame, eta, roughness, rnd_w, rec);
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYCLEARCOAT) {  // disney_clearcoat.inl:68-106
        if (dot(vertex.geometry_normal, dir_in) < 0) return true;
        Frame frame = frame_to_dir_in(vertex, dir_in);
        Vector3 half_vector = to_world(frame, sample_clearcoat_half(tex1(s, m, 0, vertex), rnd_uv));
        rec = {normalize(-dir_in + 2 * dot(dir_in, half_vector) * half_vector), Real(0), Real(1)};
        valid = true;
        return true;
    }
    if (m.kind == LJ_MAT_DISNEYBSDF) {  // disney_bsdf.inl:374-572
        Real specular_transmission = tex1(s, m, 1, vertex), metallic = tex1(s, m, 2, vertex), anisotropic = tex1(s, m, 7, vertex);
        Real clearcoat = tex1(s, m, 10, vertex), clearcoat_gloss = tex1(s, m, 11, vertex);
        Real eta = dot(vertex.geometry_normal, dir_in) > 0 ? m.eta : 1 / m.eta;
        Real diffuse_weight = (Real(1) - metallic) * (Real(1) - specular_transmission);
        Real metal_weight = (Real(1) - specular_transmission * (Real(1) - metallic));
        Real glass_weight = (Real(1) - metallic) * specular_transmission;
        Real clearcoat_weight = Real(0.25) * clearcoat;
        if (dot(vertex.geometry_normal, dir_in) < 0) {
            diffuse_weight = metal_weight = clearcoat_weight = 0;
            if (glass_weight > 0) glass_weight = 1;
            else { rec = {Vector3{0, 0, 0}, Real(0), Real(1)}; valid = true; return true; }  // a zero-direction record, NOT nullopt (:418-420)
        }
        Real weight_total = diffuse_weight + metal_weight + glass_weight + clearcoat_weight;
        diffuse_weight /= weight_total; metal_weight /= weight_total; glass_weight /= weight_total; clearcoat_weight /= weight_total;
        Real rand = rnd_w;
        if (rand < diffuse_weight) {
            rec = {to_world(frame_to_dir_in(vertex, dir_in), sample_cos_hemisphere(rnd_uv)), Real(0), Real(1)};
            valid = true;
        } else if (rand < diffuse_weight + metal_weight) {
            Frame frame = frame_to_dir_in(vertex, dir_in);
            Real roughness = clampr(tex1(s, m, 5, vertex), Real(0.01), Real(1));
            Real ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
            Vector3 half_vector = to_world(frame, sample_visible_normals_aniso(to_local(frame, dir_in), ax, ay, rnd_uv));
            rec = {normalize(-dir_in + 2 * dot(dir_in, half_vector) * half_vector), Real(0), roughness};
            valid = true;
        } else if (rand < diffuse_weight + metal_weight + glass_weight) {
            Frame frame = frame_two_sided(vertex, dir_in);
            Real roughness = clampr(tex1(s, m, 5, vertex), Real(0.01), Real(1));
            Real ax, ay; aniso_alphas(roughness, anisotropic, ax, ay);
            Vector3 half_vector = to_world(frame, sample_visible_normals_aniso(to_local(frame, dir_in), ax, ay, rnd_uv));
            Real rand_new = (rand - (diffuse_weight + metal_weight)) / glass_weight;
            valid = sample_dielectric_tail(dir_in, half_vector, frame, eta, roughness, rand_new, rec);
        } else {
            Frame frame = frame_to_dir_in(vertex, dir_in);
            Vector3 half_vector = to_world(frame, sample_clearcoat_half(clearcoat_gloss, rnd_uv));
            rec = {normalize(-dir_in + 2 * dot(dir_in, half_vector) * half_vector), Real(0), Real(1)};
            valid = true;
        }
        return true;
    }
    return false;
}

// ---- camera (camera.cpp:23-47, filters/*.inl)
Vector2 filter_sample(int kind, Real param, const Vector2 &rnd) {
    if (kind == LJ_FILTER_BOX) return {(Real(2) * rnd.x - Real(1)) * (param / 2), (Real(2) * rnd.y - Real(1)) * (param / 2)};  // box.inl:1-4
    if (kind == LJ_FILTER_TENT) {  // tent.inl:26-33
        Real h = param / 2;
        Real x = rnd.x < 0.5 ? h * (std::sqrt(2 * rnd.x) - 1) : h * (1 - std::sqrt(1 - 2 * (rnd.x - Real(0.5))));
        Real y = rnd.y < 0.5 ? h * (std::sqrt(2 * rnd.y) - 1) : h * (1 - std::sqrt(1 - 2 * (rnd.y - Real(0.5))));
        return {x, y};
    }
    Real r = param * std::sqrt(-2 * std::log(std::max(rnd.x, 1e-8)));  // gaussian.inl:1-7
    return {r * std::cos(2 * c_PI * rnd.y), r * std::sin(2 * c_PI * rnd.y)};
}
Ray sample_primary(const OScene &s, const Vector2 &screen_pos) {
    const LjCamera &cam = s.d.camera;
    Vector2 pixel_pos{screen_pos.x * cam.width, screen_pos.y * cam.height};
    Real dx = pixel_pos.x - std::floor(pixel_pos.x), dy = pixel_pos.y - std::floor(pixel_pos.y);
    Vector2 offset = filter_sample(cam.filter_kind, cam.filter_param, Vector2{dx, dy});
    Vector2 remapped{(std::floor(pixel_pos.x) + Real(0.5) + offset.x) / cam.width, (std::floor(pixel_pos.y) + Real(0.5) + offset.y) / cam.height};
    Vector3 pt = xform_point(s.sample_to_cam, Vector3{remapped.x, remapped.y, Real(0)});
    Vector3 dir = normalize(pt);
    return {xform_point(s.cam_to_world, Vector3{0, 0, 0}), normalize(xform_vector(s.cam_to_world, dir)), Real(0), std::numeric_limits<Real>::infinity()};
}

// ------------------------------------------------------------------ closest-hit / any-hit over the scene
// Candidate order never matters: closest = minimum (t, global primitive index).
inline void prim_closest(const OScene &s, long long g, const float o[3], const float d[3], float tnear, float tfar, Hit &best) {
    int sid = s.prim_shape[g], lp = s.prim_local[g];
    const LjShape &sh = s.shapes[sid];
    if (sh.kind == LJ_SHAPE_TRIMESH) {
        const OMesh &m = s.meshes[sid];
        const float *pf = m.Pf.data();
        float t, u, v;
        if (tri_test(o, d, tnear, best.t, pf + 3 * m.I[3 * lp], pf + 3 * m.I[3 * lp + 1], pf + 3 * m.I[3 * lp + 2], t, u, v)) {
            if (t < best.t || (t == best.t && g < best.gprim)) { best.t = t; best.u = u; best.v = v; best.shape_id = sid; best.prim_id = lp; best.gprim = g; }
        }
    } else {
        Real t;
        // The reference's callback compares against the *current* rtc_ray->tfar, which makes exact ties depend on
        // Embree's traversal order; we range-check against the ray's original tfar and let (float t, primitive id)
        // decide, so the result is traversal-order independent (differs from the reference only on exact ties).
        if (sphere_test(o, d, tnear, tfar, Vector3{sh.position[0], sh.position[1], sh.position[2]}, sh.radius, t)) {
            float tf = (float)t;
            if (tf < best.t || (tf == best.t && g < best.gprim)) { best.t = tf; best.u = 0; best.v = 0; best.shape_id = sid; best.prim_id = 0; best.gprim = g; best.t_sphere = t; }
        }
    }
}
inline bool box_hit(const BvhNode &n, const float o[3], const float inv[3], float tnear, float tfar) {
    float t0 = tnear, t1 = tfar;
    for (int k = 0; k < 3; k++) {
        float a = (n.lo[k] - o[k]) * inv[k], b = (n.hi[k] - o[k]) * inv[k];
        float lo = fminf(a, b), hi = fmaxf(a, b);
        if (lo != lo) lo = -INFINITY;  // NaN from 0*inf: treat the slab as unbounded on that side
        if (hi != hi) hi = INFINITY;
        t0 = fmaxf(t0, lo); t1 = fminf(t1, hi);
    }
    return t0 <= t1 * 1.0000005f;
}
// ANY: stop at the first accepted primitive (occluded(): "a hit exists in [tnear, tfar]"; which one is irrelevant).
// The child on the ray's side of the split plane is visited first; since the closest hit is min (t, primitive id), the
// visiting order changes the work done, never the result.
template <bool ANY>
Hit scene_trace(const OScene &s, const float o[3], const float d[3], float tnear, float tfar) {
    Hit best{tfar, 0, 0, -1, -1, (long long)1 << 62, 0.0};
    if (!s.use_bvh) {
        for (long long g = 0; g < (long long)s.prim_shape.size(); g++) { prim_closest(s, g, o, d, tnear, tfar, best); if (ANY && best.shape_id >= 0) break; }
    } else {
        float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        int stack[128], sp = 0; stack[sp++] = 0;
        while (sp) {
            const BvhNode &n = s.nodes[stack[--sp]];
            if (!box_hit(n, o, inv, tnear, best.t)) continue;
            if (n.count > 0) {
                for (int i = 0; i < n.count; i++) prim_closest(s, s.prim_order[n.first + i], o, d, tnear, tfar, best);
                if (ANY && best.shape_id >= 0) break;
            } else if (d[n.axis] >= 0.0f) { stack[sp++] = n.right; stack[sp++] = n.left; }
            else { stack[sp++] = n.left; stack[sp++] = n.right; }
        }
    }
    return best;
}
Hit scene_intersect(const OScene &s, const float o[3], const float d[3], float tnear, float tfar) { return scene_trace<false>(s, o, d, tnear, tfar); }
bool scene_occluded(const OScene &s, const float o[3], const float d[3], float tnear, float tfar) {
    return scene_trace<true>(s, o, d, tnear, tfar).shape_id >= 0;
}

// Triangle geometry normal the way Embree reports it: (p1-p0)x(p2-p0) on the float vertices.
Vector3 tri_Ng(const OScene &s, const Hit &h) {
    const OMesh &m = s.meshes[h.shape_id];
    const float *a = &m.Pf[3 * m.I[3 * h.prim_id]], *b = &m.Pf[3 * m.I[3 * h.prim_id + 1]], *c = &m.Pf[3 * m.I[3 * h.prim_id + 2]];
    float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    return {(Real)(e1[1] * e2[2] - e1[2] * e2[1]), (Real)(e1[2] * e2[0] - e1[0] * e2[2]), (Real)(e1[0] * e2[1] - e1[1] * e2[0])};
}

struct Counters { uint64_t samples = 0, bounces = 0, rays_closest = 0, rays_shadow = 0; };

// intersect() (intersection.cpp:7-65) on our traversal
bool intersect(const OScene &s, const Ray &ray, const RayDifferential &rd, PathVertex &out, Counters *cnt) {
    float o[3] = {(float)ray.org.x, (float)ray.org.y, (float)ray.org.z}, d[3] = {(float)ray.dir.x, (float)ray.dir.y, (float)ray.dir.z};
    if (cnt) cnt->rays_closest++;
    Hit h = scene_intersect(s, o, d, (float)ray.tnear, (float)ray.tfar);
    if (h.shape_id < 0) return false;
    const LjShape &sh = s.shapes[h.shape_id];
    Vector3 Ng; float u = h.u, v = h.v;
    if (sh.kind == LJ_SHAPE_TRIMESH) Ng = tri_Ng(s, h);
    else {  // sphere.inl:85-99: Ng, u, v are computed in double from the double t, then narrowed to float fields
        Real t = h.t_sphere;
        Vector3 org{o[0], o[1], o[2]}, dir{d[0], d[1], d[2]}, c{sh.position[0], sh.position[1], sh.position[2]};
        Vector3 p = org + t * dir;
        Vector3 gn = p - c;
        Ng = {(Real)(float)gn.x, (Real)(float)gn.y, (Real)(float)gn.z};
        Vector3 cart = gn / sh.radius;
        Real elevation = std::acos(clampr(cart.y, Real(-1), Real(1))), azimuth = std::atan2(cart.z, cart.x);
        u = (float)(azimuth / c_TWOPI); v = (float)(elevation / c_PI);
    }
    out = make_vertex(s, ray, rd, h.shape_id, h.prim_id, h.t, u, v, Ng);
    return true;
}
bool occluded(const OScene &s, const Ray &ray, Counters *cnt) {  // intersection.cpp:67-85
    float o[3] = {(float)ray.org.x, (float)ray.org.y, (float)ray.org.z}, d[3] = {(float)ray.dir.x, (float)ray.dir.y, (float)ray.dir.z};
    if (cnt) cnt->rays_shadow++;
    return scene_occluded(s, o, d, (float)ray.tnear, (float)ray.tfar);
}
Spectrum vertex_emission(const OScene &s, const PathVertex &v, const Vector3 &view_dir) {  // intersection.cpp:87-98
    int light_id = s.shapes[v.shape_id].area_light_id;
    return light_emission(s, light_id, view_dir, v.uv_screen_size, PointAndNormal{v.position, v.geometry_normal});
}

// ------------------------------------------------------------------ path_tracing (path_tracing.h:7-325)
// status: 0 ok, 1 unsupported material encountered
Spectrum path_tracing(const OScene &scene, int x, int y, pcg32_state &rng, int max_depth_override, bool use_override, Counters *cnt, int *status) {
    const LjCamera &cam = scene.d.camera;
    int w = cam.width, h = cam.height;
    // g++ evaluates the two constructor arguments right-to-left: the FIRST draw jitters y, the SECOND x
    Real jy = next_pcg32_real(rng);
    Real jx = next_pcg32_real(rng);
    Vector2 screen_pos{(x + jx) / w, (y + jy) / h};
    Ray ray = sample_primary(scene, screen_pos);
    RayDifferential ray_diff{Real(0), Real(0.25) / std::max(w, h)};  // ray.h:35-37
    if (cnt) cnt->samples++;
    PathVertex vertex;
    if (!intersect(scene, ray, ray_diff, vertex, cnt)) {
        if (scene.d.envmap_light_id != -1)
            return light_emission(scene, scene.d.envmap_light_id, -ray.dir, ray_diff.spread, PointAndNormal{});
        return {0, 0, 0};
    }
    Spectrum radiance{0, 0, 0};
    Spectrum current_path_throughput{1, 1, 1};
    Real eta_scale = Real(1);
    if (scene.shapes[vertex.shape_id].area_light_id >= 0) radiance += current_path_throughput * vertex_emission(scene, vertex, -ray.dir);
    int max_depth = use_override ? max_depth_override : scene.d.options.max_depth;
    const Real eps = shadow_epsilon(scene);
    for (int num_vertices = 3; max_depth == -1 || num_vertices <= max_depth + 1; num_vertices++) {
        if (cnt) cnt->bounces++;
        const LjMaterial &mat = scene.materials[vertex.material_id];
        Vector2 light_uv; light_uv.x = next_pcg32_real(rng); light_uv.y = next_pcg32_real(rng);
        Real light_w = next_pcg32_real(rng);
        Real shape_w = next_pcg32_real(rng);
        int light_id = sample_1d(scene.light_dist, light_w);
        const LjLight &light = scene.lights[light_id];
        PointAndNormal point_on_light = sample_point_on_light(scene, light_id, vertex.position, light_uv, shape_w);
        Spectrum C1{0, 0, 0}; Real w1 = 0;
        {
            Real G = 0; Vector3 dir_light;
            if (light.kind != LJ_LIGHT_ENVMAP) {
                dir_light = normalize(point_on_light.position - vertex.position);
                Ray shadow_ray{vertex.position, dir_light, eps, (1 - eps) * distance(point_on_light.position, vertex.position)};
                if (!occluded(scene, shadow_ray, cnt))
                    G = std::max(-dot(dir_light, point_on_light.normal), Real(0)) / distance_squared(point_on_light.position, vertex.position);
            } else {
                dir_light = -point_on_light.normal;
                Ray shadow_ray{vertex.position, dir_light, eps, std::numeric_limits<Real>::infinity()};
                if (!occluded(scene, shadow_ray, cnt)) G = 1;
            }
            Real p1 = scene.light_dist.pmf[light_id] * pdf_point_on_light(scene, light_id, point_on_light, vertex.position);
            if (G > 0 && p1 > 0) {
                Vector3 dir_view = -ray.dir;
                Spectrum f; if (!bsdf_eval(&scene, mat, dir_view, dir_light, vertex, f)) { *status = 1; return {0, 0, 0}; }
                Spectrum L = light_emission(scene, light_id, -dir_light, Real(0), point_on_light);
                C1 = G * f * L;
                Real p2; bsdf_pdf(&scene, mat, dir_view, dir_light, vertex, p2);
                p2 *= G;
                w1 = (p1 * p1) / (p1 * p1 + p2 * p2);
                C1 = C1 / p1;
            }
        }
        radiance += current_path_throughput * C1 * w1;

        Vector3 dir_view = -ray.dir;
        Vector2 bsdf_rnd_param_uv; bsdf_rnd_param_uv.x = next_pcg32_real(rng); bsdf_rnd_param_uv.y = next_pcg32_real(rng);
        Real bsdf_rnd_param_w = next_pcg32_real(rng);
        bool valid; BSDFSampleRecord bsdf_sample_rec;
        if (!bsdf_sample(&scene, mat, dir_view, vertex, bsdf_rnd_param_uv, bsdf_rnd_param_w, valid, bsdf_sample_rec)) { *status = 1; return {0, 0, 0}; }
        if (!valid) break;
        Vector3 dir_bsdf = bsdf_sample_rec.dir_out;
        if (bsdf_sample_rec.eta == 0) ray_diff.spread = rd_reflect(ray_diff, vertex.mean_curvature, bsdf_sample_rec.roughness);
        else { ray_diff.spread = rd_refract(ray_diff, vertex.mean_curvature, bsdf_sample_rec.eta, bsdf_sample_rec.roughness); eta_scale /= (bsdf_sample_rec.eta * bsdf_sample_rec.eta); }
        Ray bsdf_ray{vertex.position, dir_bsdf, eps, std::numeric_limits<Real>::infinity()};
        PathVertex bsdf_vertex;
        bool hit = intersect(scene, bsdf_ray, RayDifferential{}, bsdf_vertex, cnt);  // default RayDifferential{0,0} (path_tracing.h:237)
        Real G;
        if (hit) G = std::fabs(dot(dir_bsdf, bsdf_vertex.geometry_normal)) / distance_squared(bsdf_vertex.position, vertex.position);
        else G = 1;
        Spectrum f; bsdf_eval(&scene, mat, dir_view, dir_bsdf, vertex, f);
        Real p2; bsdf_pdf(&scene, mat, dir_view, dir_bsdf, vertex, p2);
        if (p2 <= 0) break;
        p2 *= G;
        if (hit && scene.shapes[bsdf_vertex.shape_id].area_light_id >= 0) {
            Spectrum L = vertex_emission(scene, bsdf_vertex, -dir_bsdf);
            Spectrum C2 = G * f * L;
            int lid = scene.shapes[bsdf_vertex.shape_id].area_light_id;
            PointAndNormal light_point{bsdf_vertex.position, bsdf_vertex.geometry_normal};
            Real p1 = scene.light_dist.pmf[lid] * pdf_point_on_light(scene, lid, light_point, vertex.position);
            Real w2 = (p2 * p2) / (p1 * p1 + p2 * p2);
            C2 = C2 / p2;
            radiance += current_path_throughput * C2 * w2;
        } else if (!hit && scene.d.envmap_light_id != -1) {
            int lid = scene.d.envmap_light_id;
            Spectrum L = light_emission(scene, lid, -dir_bsdf, ray_diff.spread, PointAndNormal{});
            Spectrum C2 = G * f * L;
            PointAndNormal light_point{Vector3{0, 0, 0}, -dir_bsdf};
            Real p1 = scene.light_dist.pmf[lid] * pdf_point_on_light(scene, lid, light_point, vertex.position);
            Real w2 = (p2 * p2) / (p1 * p1 + p2 * p2);
            C2 = C2 / p2;
            radiance += current_path_throughput * C2 * w2;
        }
        if (!hit) break;
        Real rr_prob = 1;
        if (num_vertices - 1 >= scene.d.options.rr_depth) {
            rr_prob = std::min(vmax((1 / eta_scale) * current_path_throughput), Real(0.95));
            if (next_pcg32_real(rng) > rr_prob) break;
        }
        ray = bsdf_ray; vertex = bsdf_vertex;
        current_path_throughput = current_path_throughput * (G * f) / (p2 * rr_prob);
    }
    return radiance;
}

// ------------------------------------------------------------------ scene construction (scene.cpp:30-52)
void build_bvh(OScene &s) {
    size_t n = s.prim_shape.size();
    s.prim_order.resize(n);
    std::vector<float> lo(3 * n), hi(3 * n), ctr(3 * n);
    for (size_t g = 0; g < n; g++) {
        s.prim_order[g] = (int)g;
        int sid = s.prim_shape[g], lp = s.prim_local[g];
        const LjShape &sh = s.shapes[sid];
        for (int k = 0; k < 3; k++) {
            float l, h;
            if (sh.kind == LJ_SHAPE_TRIMESH) {
                const OMesh &m = s.meshes[sid];
                float a = m.Pf[3 * m.I[3 * lp] + k], b = m.Pf[3 * m.I[3 * lp + 1] + k], c = m.Pf[3 * m.I[3 * lp + 2] + k];
                l = fminf(a, fminf(b, c)); h = fmaxf(a, fmaxf(b, c));
            } else { l = (float)(sh.position[k] - sh.radius); h = (float)(sh.position[k] + sh.radius); }
            float pad = 1e-5f * (fabsf(l) + fabsf(h)) + 1e-7f * (h - l) + 1e-30f;  // conservative: the box test must never cull a hit the primitive test accepts
            lo[3 * g + k] = l - pad; hi[3 * g + k] = h + pad; ctr[3 * g + k] = 0.5f * (l + h);
        }
    }
    struct Task { int node, first, count; };
    s.nodes.clear(); s.nodes.push_back(BvhNode{});
    std::vector<Task> stack{{0, 0, (int)n}};
    while (!stack.empty()) {
        Task t = stack.back(); stack.pop_back();
        BvhNode nd{}; for (int k = 0; k < 3; k++) { nd.lo[k] = INFINITY; nd.hi[k] = -INFINITY; }
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < t.count; i++) {
            int g = s.prim_order[t.first + i];
            for (int k = 0; k < 3; k++) { nd.lo[k] = fminf(nd.lo[k], lo[3 * g + k]); nd.hi[k] = fmaxf(nd.hi[k], hi[3 * g + k]); clo[k] = fminf(clo[k], ctr[3 * g + k]); chi[k] = fmaxf(chi[k], ctr[3 * g + k]); }
        }
        int axis = 0; for (int k = 1; k < 3; k++) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
        if (t.count <= 4 || !(chi[axis] > clo[axis])) { nd.first = t.first; nd.count = t.count; nd.left = nd.right = -1; s.nodes[t.node] = nd; continue; }
        int mid = t.first + t.count / 2;
        std::nth_element(s.prim_order.begin() + t.first, s.prim_order.begin() + mid, s.prim_order.begin() + t.first + t.count,
                         [&](int a, int b) { return ctr[3 * a + axis] < ctr[3 * b + axis]; });
        nd.count = 0; nd.first = 0; nd.axis = axis; nd.left = (int)s.nodes.size(); nd.right = nd.left + 1;
        s.nodes.push_back(BvhNode{}); s.nodes.push_back(BvhNode{});
        s.nodes[t.node] = nd;
        stack.push_back({nd.left, t.first, mid - t.first}); stack.push_back({nd.right, mid, t.first + t.count - mid});
    }
}

OScene *scene_create(const LjSceneDesc *d) {
    OScene *s = new OScene();
    s->d = *d;
    s->shapes.assign(d->shapes, d->shapes + d->n_shapes);
    s->materials.assign(d->materials, d->materials + d->n_materials);
    s->lights.assign(d->lights, d->lights + d->n_lights);
    memcpy(s->sample_to_cam.m, d->camera.sample_to_cam, sizeof(Real) * 16);
    memcpy(s->cam_to_world.m, d->camera.cam_to_world, sizeof(Real) * 16);
    s->meshes.resize(d->n_shapes);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    long long g = 0;
    for (int i = 0; i < d->n_shapes; i++) {
        const LjShape &sh = d->shapes[i];
        if (sh.kind == LJ_SHAPE_TRIMESH) {
            OMesh &m = s->meshes[i];
            m.P = d->positions + 3 * sh.first_vertex; m.N = d->normals + 3 * sh.first_vertex; m.UV = d->uvs + 2 * sh.first_vertex;
            m.I = d->indices + 3 * sh.first_triangle; m.nv = sh.n_vertices; m.nt = sh.n_triangles; m.has_n = sh.has_normals; m.has_uv = sh.has_uvs;
            m.Pf.resize(3 * m.nv);
            for (int64_t k = 0; k < 3 * m.nv; k++) m.Pf[k] = (float)m.P[k];  // triangle_mesh.inl:11-14
            for (int64_t v = 0; v < m.nv; v++) for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], m.Pf[3 * v + k]); hi[k] = fmaxf(hi[k], m.Pf[3 * v + k]); }
            // init_sampling_dist (triangle_mesh.inl:48-63)
            std::vector<Real> areas(m.nt); Real total = 0;
            for (int64_t t = 0; t < m.nt; t++) {
                Vector3 v0 = mesh_p(m, m.I[3 * t]), v1 = mesh_p(m, m.I[3 * t + 1]), v2 = mesh_p(m, m.I[3 * t + 2]);
                areas[t] = length(cross(v1 - v0, v2 - v0)) / 2; total += areas[t];
            }
            m.tri_sampler = make_table_dist_1d(areas); m.total_area = total;
            m.gprim0 = g;
            for (int64_t t = 0; t < m.nt; t++) { s->prim_shape.push_back(i); s->prim_local.push_back((int)t); }
            g += m.nt;
        } else {
            for (int k = 0; k < 3; k++) {  // sphere.inl:1-10 into float RTCBounds
                lo[k] = fminf(lo[k], (float)(sh.position[k] - sh.radius)); hi[k] = fmaxf(hi[k], (float)(sh.position[k] + sh.radius));
            }
            s->prim_shape.push_back(i); s->prim_local.push_back(0); g++;
        }
    }
    Vector3 lb{lo[0], lo[1], lo[2]}, ub{hi[0], hi[1], hi[2]};  // scene.cpp:30-34
    s->bounds_radius = distance(ub, lb) / 2; s->bounds_center = (lb + ub) / Real(2);
    for (int i = 0; i < d->n_images3; i++) s->mips3.push_back(make_mipmap(d->images3[i]));
    for (int i = 0; i < d->n_images1; i++) s->mips1.push_back(make_mipmap(d->images1[i]));
    s->env_dist.resize(d->n_lights);
    for (int i = 0; i < d->n_lights; i++) {  // envmap.inl:75-98
        const LjLight &l = d->lights[i];
        if (l.kind != LJ_LIGHT_ENVMAP || l.values.kind != LJ_TEX_IMAGE) continue;
        const Mip &mm = s->mips3[l.values.texture_id];
        int w = mm.w[0], h = mm.h[0];
        std::vector<Real> f((size_t)w * h); size_t k = 0;
        for (int y = 0; y < h; y++) {
            Real v = (y + Real(0.5)) / Real(h), sin_elevation = std::sin(c_PI * v);
            for (int x = 0; x < w; x++) { Real u = (x + Real(0.5)) / Real(w); f[k++] = luminance(mip_lookup_level(mm, u, v, 0)) * sin_elevation; }
        }
        s->env_dist[i] = make_table_dist_2d(f, w, h);
    }
    std::vector<Real> power(d->n_lights);  // scene.cpp:47-52
    for (int i = 0; i < d->n_lights; i++) power[i] = light_power(*s, i);
    s->light_dist = make_table_dist_1d(power);
    s->use_bvh = s->prim_shape.size() > 64;
    if (s->use_bvh) build_bvh(*s);
    return s;
}

} // namespace

// =================================================================== C entry points for the tests (ctypes)
// ------------------------------------------------------------------ participating media
// lajolla.h:63-71: the reference's own min / max templates.  With a NaN first argument they return the SECOND one, which
// is what keeps Russian roulette alive (probability 0.95) when a throughput has degenerated to 0 / 0 after a chain of
// null collisions underflowed both the transmittance and its pdf — std::min would return the NaN and never terminate.
inline Real ref_min(Real a, Real b) { return a < b ? a : b; }
inline Real ref_max(Real a, Real b) { return a > b ? a : b; }
inline Real ref_vmax(const Vector3 &v) { return ref_max(ref_max(v.x, v.y), v.z); }
inline Vector3 vmul(const Vector3 &a, const Vector3 &b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vector3 vdiv(const Vector3 &a, const Vector3 &b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline Vector3 vexp(const Vector3 &a) { return {std::exp(a.x), std::exp(a.y), std::exp(a.z)}; }
inline Real vavg(const Vector3 &a) { return (a.x + a.y + a.z) / Real(3); }   // vector.h:259-262
inline Real vget(const Vector3 &a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
inline Vector3 v3of(const double *d) { return {d[0], d[1], d[2]}; }

// lookup(VolumeSpectrum, p) (volume.h:39-81): trilinear interpolation inside [p_min, p_max], zero outside
Vector3 volume_lookup(const LjVolume &v, const Vector3 &p) {
    if (v.kind == LJ_VOLUME_CONSTANT) return v3of(v.value);
    Vector3 pn = vdiv(p - v3of(v.p_min), v3of(v.p_max) - v3of(v.p_min));
    if (pn.x < 0 || pn.x > 1 || pn.y < 0 || pn.y > 1 || pn.z < 0 || pn.z > 1) return {0, 0, 0};
    const int rx = v.resolution[0], ry = v.resolution[1], rz = v.resolution[2];
    pn.x *= Real(rx - 1); pn.y *= Real(ry - 1); pn.z *= Real(rz - 1);
    auto clampi = [](int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); };
    int x0 = clampi(int(pn.x), 0, rx - 1), y0 = clampi(int(pn.y), 0, ry - 1), z0 = clampi(int(pn.z), 0, rz - 1);
    int x1 = clampi(x0 + 1, 0, rx - 1), y1 = clampi(y0 + 1, 0, ry - 1), z1 = clampi(z0 + 1, 0, rz - 1);
    Real dx = pn.x - x0, dy = pn.y - y0, dz = pn.z - z0;
    auto at = [&](int x, int y, int z) { const float *d = v.data + 3 * ((size_t)(z * ry + y) * rx + x); return Vector3{d[0], d[1], d[2]}; };
    return v.scale * (at(x0, y0, z0) * ((1 - dx) * (1 - dy) * (1 - dz)) + at(x1, y0, z0) * (dx * (1 - dy) * (1 - dz)) +
                      at(x0, y1, z0) * ((1 - dx) * dy * (1 - dz)) + at(x1, y1, z0) * (dx * dy * (1 - dz)) +
                      at(x0, y0, z1) * ((1 - dx) * (1 - dy) * dz) + at(x1, y0, z1) * (dx * (1 - dy) * dz) +
                      at(x0, y1, z1) * ((1 - dx) * dy * dz) + at(x1, y1, z1) * (dx * dy * dz));
}
// intersect(Volume, ray) (volume.h:118-144): slab test of the grid's box over [0, ray.tfar]
bool volume_intersect(const LjVolume &v, const Ray &ray) {
    if (v.kind == LJ_VOLUME_CONSTANT) return true;
    Real t0 = 0, t1 = ray.tfar;
    const Real o[3] = {ray.org.x, ray.org.y, ray.org.z}, d[3] = {ray.dir.x, ray.dir.y, ray.dir.z};
    for (int i = 0; i < 3; i++) {
        Real tn = (v.p_min[i] - o[i]) / d[i], tf = (v.p_max[i] - o[i]) / d[i];
        if (tn > tf) std::swap(tn, tf);
        t0 = tn > t0 ? tn : t0; t1 = tf < t1 ? tf : t1;
        if (t0 > t1) return false;
    }
    return true;
}
Vector3 volume_max(const LjVolume &v) { return v.kind == LJ_VOLUME_CONSTANT ? v3of(v.value) : v.scale * v3of(v.max_data); }  // volume.h:83-97
// medium.cpp:27-37 with media/homogeneous.inl, media/heterogeneous.inl
Vector3 get_majorant(const LjMedium &m, const Ray &ray) {
    if (m.kind == LJ_MEDIUM_HOMOGENEOUS) return v3of(m.sigma_a) + v3of(m.sigma_s);
    return volume_intersect(m.density, ray) ? volume_max(m.density) : Vector3{0, 0, 0};
}
Vector3 get_sigma_s(const LjMedium &m, const Vector3 &p) {
    if (m.kind == LJ_MEDIUM_HOMOGENEOUS) return v3of(m.sigma_s);
    return vmul(volume_lookup(m.density, p), volume_lookup(m.albedo, p));
}
Vector3 get_sigma_a(const LjMedium &m, const Vector3 &p) {
    if (m.kind == LJ_MEDIUM_HOMOGENEOUS) return v3of(m.sigma_a);
    Vector3 a = volume_lookup(m.albedo, p);
    return vmul(volume_lookup(m.density, p), Vector3{1 - a.x, 1 - a.y, 1 - a.z});
}
// phase_functions/isotropic.inl, henyeygreenstein.inl
Real phase_eval(const LjMedium &m, const Vector3 &dir_in, const Vector3 &dir_out) {   // == pdf_sample_phase
    if (m.phase_kind == LJ_PHASE_ISOTROPIC) return c_INVFOURPI;
    const Real g = m.g;
    return c_INVFOURPI * (1 - g * g) / std::pow(1 + g * g + 2 * g * dot(dir_in, dir_out), Real(3) / Real(2));
}
Vector3 phase_sample(const LjMedium &m, const Vector3 &dir_in, const Vector2 &rnd) {
    const Real g = m.g;
    if (m.phase_kind == LJ_PHASE_ISOTROPIC || std::fabs(g) < Real(1e-3)) {
        Real z = 1 - 2 * rnd.x, r = std::sqrt(std::fmax(Real(0), 1 - z * z)), phi = 2 * c_PI * rnd.y;
        return {r * std::cos(phi), r * std::sin(phi), z};
    }
    Real tmp = (g * g - 1) / (2 * rnd.x * g - (g + 1));
    Real cos_el = (tmp * tmp - (1 + g * g)) / (2 * g);
    Real sin_el = std::sqrt(std::max(1 - cos_el * cos_el, Real(0)));
    Real az = 2 * c_PI * rnd.y;
    return to_world(make_frame(dir_in), Vector3{sin_el * std::cos(az), sin_el * std::sin(az), cos_el});
}

// update_medium (vol_path_tracing.h:149-163)
inline int update_medium(const OScene &s, const PathVertex &v, const Ray &ray, int medium) {
    const LjShape &sh = s.shapes[v.shape_id];
    if (sh.interior_medium_id != sh.exterior_medium_id) medium = dot(ray.dir, v.geometry_normal) > 0 ? sh.exterior_medium_id : sh.interior_medium_id;
    return medium;
}

// next_event_estimation_final (vol_path_tracing.h:299-494).  The reference dereferences an empty optional at :344
// (`shadow_vertex = *shadow_vertex_`); every later use of that value is guarded by the optional, so it is skipped here.
Spectrum vol_nee(const OScene &scene, pcg32_state &rng, Vector3 p, int current_medium, int bounces, const Vector3 &dir_view,
                 bool is_surface, const PathVertex &vertex, int max_depth, Counters *cnt, int *status) {
    Vector2 light_uv; light_uv.x = next_pcg32_real(rng); light_uv.y = next_pcg32_real(rng);
    Real light_w = next_pcg32_real(rng);
    Real shape_w = next_pcg32_real(rng);
    int light_id = sample_1d(scene.light_dist, light_w);
    PointAndNormal pl = sample_point_on_light(scene, light_id, p, light_uv, shape_w);
    Vector3 dir_light = normalize(pl.position - p);
    const Vector3 p_prime = pl.position, p_origin = p;
    const Real eps = shadow_epsilon(scene);
    int shadow_medium = current_medium, shadow_bounces = 0;
    Spectrum T{1, 1, 1}, p_trans_nee{1, 1, 1}, p_trans_dir{1, 1, 1};
    for (;;) {
        Ray shadow_ray{p, dir_light, eps, (1 - eps) * distance(p, p_prime)};
        PathVertex sv; const bool hit = intersect(scene, shadow_ray, RayDifferential{0, 0}, sv, cnt);
        Real next_t = distance(p, p_prime);
        if (hit) next_t = distance(p, sv.position);
        if (shadow_medium != -1) {
            const LjMedium &med = scene.d.media[shadow_medium];
            Vector3 majorant = get_majorant(med, shadow_ray);
            Real u = next_pcg32_real(rng);
            int channel = std::min(std::max(int(u * 3), 0), 2);
            Real accum_t = 0; int iteration = 0;
            for (;;) {
                if (vget(majorant, channel) <= 0) break;
                if (iteration >= scene.d.options.max_null_collisions) break;
                Real t = -std::log(1 - next_pcg32_real(rng)) / vget(majorant, channel);
                Real dt = next_t - accum_t;
                accum_t = std::min(accum_t + t, next_t);
                if (t < dt) {
                    Vector3 pos = p + dir_light * accum_t;
                    Vector3 sigma_t = get_sigma_s(med, pos) + get_sigma_a(med, pos);
                    Vector3 ratio = vdiv(sigma_t, majorant);
                    Vector3 sigma_n = vmul(majorant, Vector3{1 - ratio.x, 1 - ratio.y, 1 - ratio.z});
                    Vector3 e = vexp(-(majorant * t));
                    Real mx = vmax(majorant);
                    T = vmul(T, vmul(e, sigma_n) / mx);
                    p_trans_nee = vmul(p_trans_nee, vmul(e, majorant) / mx);
                    p_trans_dir = vmul(p_trans_dir, vmul(vmul(e, majorant), Vector3{1 - ratio.x, 1 - ratio.y, 1 - ratio.z}) / mx);
                    if (vmax(T) <= 0) break;
                } else {
                    Vector3 e = vexp(-(majorant * dt));
                    T = vmul(T, e); p_trans_nee = vmul(p_trans_nee, e); p_trans_dir = vmul(p_trans_dir, e);
                    break;
                }
                iteration++;
            }
        }
        if (!hit) break;
        if (sv.material_id >= 0) return {0, 0, 0};   // an opaque surface blocks the light
        shadow_bounces++;                             // an index-matched surface: pass through, one more connection vertex
        if (max_depth != -1 && bounces + shadow_bounces >= max_depth) return {0, 0, 0};
        shadow_medium = update_medium(scene, sv, shadow_ray, shadow_medium);
        p = p + next_t * dir_light;
    }
    if (!(vmax(T) > 0)) return {0, 0, 0};
    Spectrum Le = light_emission(scene, light_id, -dir_light, Real(0), pl);
    Real jacobian = std::max(-dot(dir_light, pl.normal), Real(0)) / distance_squared(p_origin, p_prime);
    Spectrum pdf_nee = (scene.light_dist.pmf[light_id] * pdf_point_on_light(scene, light_id, pl, p_origin)) * p_trans_nee;
    Spectrum f, pdf_dir;
    if (is_surface) {
        const LjMaterial &mat = scene.materials[vertex.material_id];
        if (!bsdf_eval(&scene, mat, dir_view, dir_light, vertex, f)) { *status = 1; return {0, 0, 0}; }
        Real pdf_bsdf; if (!bsdf_pdf(&scene, mat, dir_view, dir_light, vertex, pdf_bsdf)) { *status = 1; return {0, 0, 0}; }
        if (pdf_bsdf <= 0) return {0, 0, 0};
        pdf_dir = (pdf_bsdf * jacobian) * p_trans_dir;
    } else {
        const LjMedium &med = scene.d.media[current_medium];
        next_pcg32_real(rng); next_pcg32_real(rng);   // phase_uv: drawn and never used (vol_path_tracing.h:475)
        Real ph = phase_eval(med, dir_view, dir_light);
        f = {ph, ph, ph};
        pdf_dir = (ph * jacobian) * p_trans_dir;
    }
    Spectrum contrib = vmul(vmul(T, f), Le) * jacobian / vavg(pdf_nee);
    Spectrum n2 = vmul(pdf_nee, pdf_nee), d2 = vmul(pdf_dir, pdf_dir);
    Spectrum w = vdiv(n2, n2 + d2);
    return vmul(contrib, w);
}

// vol_path_tracing_1 (vol_path_tracing.h:6-41): absorption only, a directly visible emitter seen through the exterior medium of the
// surface it sits on.  (The primary ray keeps sample_primary's tnear of 0; only the final version moves it to the epsilon.)
Spectrum vol_path_tracing_1(const OScene &scene, int x, int y, pcg32_state &rng, Counters *cnt) {
    const LjCamera &cam = scene.d.camera;
    Real jy = next_pcg32_real(rng);   // g++ evaluates the constructor arguments right to left
    Real jx = next_pcg32_real(rng);
    Ray ray = sample_primary(scene, Vector2{(x + jx) / cam.width, (y + jy) / cam.height});
    if (cnt) cnt->samples++;
    PathVertex vertex;
    if (!intersect(scene, ray, RayDifferential{0, 0}, vertex, cnt)) return {0, 0, 0};
    const int exterior = scene.shapes[vertex.shape_id].exterior_medium_id;   // (PathVertex::exterior_medium_id, intersection.cpp:60)
    if (exterior == -1) return {0, 0, 0};
    const LjMedium &med = scene.d.media[exterior];
    const Real t_hit = distance(vertex.position, ray.org);
    const Vector3 sigma_a = get_sigma_a(med, vertex.position);
    const Spectrum transmittance = vexp(-(sigma_a * t_hit));
    Spectrum Le{0, 0, 0};
    if (scene.shapes[vertex.shape_id].area_light_id >= 0) Le = vertex_emission(scene, vertex, -ray.dir);
    return vmul(transmittance, Le);
}

// vol_path_tracing_2 (vol_path_tracing.h:46-147): one monochromatic homogeneous medium, single scattering; free-flight sampling on
// the red channel.  The reference reads `*vertex_` before it knows the ray hit anything (:59); for a ray that leaves the scene the
// cross sections are looked up at the ray origin here (a homogeneous medium — the only kind this version is meant for — has no
// position dependence); the `t >= t_hit` arm is reached by such a ray only when sigma_t.x is 0 there, and returns zero.
Spectrum vol_path_tracing_2(const OScene &scene, int x, int y, pcg32_state &rng, Counters *cnt) {
    const LjCamera &cam = scene.d.camera;
    Real jy = next_pcg32_real(rng);
    Real jx = next_pcg32_real(rng);
    Ray ray = sample_primary(scene, Vector2{(x + jx) / cam.width, (y + jy) / cam.height});
    if (cnt) cnt->samples++;
    PathVertex vertex;
    const bool hit = intersect(scene, ray, RayDifferential{0, 0}, vertex, cnt);
    const int medium_id = hit ? scene.shapes[vertex.shape_id].exterior_medium_id : cam.medium_id;
    if (medium_id < 0) return {0, 0, 0};   // (scene.media[-1] in the reference: undefined; nothing sensible to restate)
    const Real t_hit = hit ? distance(vertex.position, ray.org) : std::numeric_limits<Real>::infinity();
    const LjMedium &med = scene.d.media[medium_id];
    const Vector3 at = hit ? vertex.position : ray.org;
    const Vector3 sigma_s = get_sigma_s(med, at), sigma_a = get_sigma_a(med, at), sigma_t = sigma_s + sigma_a;
    const Real u = next_pcg32_real(rng);
    const Real t = -std::log(1 - u) / sigma_t.x;
    if (t < t_hit) {
        const Spectrum transmittance = vexp(-(sigma_t * t)), trans_pdf = vmul(transmittance, sigma_t);
        const Vector3 p = ray.org + ray.dir * t;
        Vector2 light_uv; light_uv.x = next_pcg32_real(rng); light_uv.y = next_pcg32_real(rng);
        const Real light_w = next_pcg32_real(rng);
        const Real shape_w = next_pcg32_real(rng);
        const int light_id = sample_1d(scene.light_dist, light_w);
        const PointAndNormal pl = sample_point_on_light(scene, light_id, p, light_uv, shape_w);
        const Vector3 dir_light = normalize(pl.position - p);
        const Real rho = phase_eval(med, -ray.dir, dir_light);
        const Spectrum Le = light_emission(scene, light_id, -dir_light, Real(0), pl);
        const Real dist = distance(p, pl.position);
        const Spectrum exp_term = vexp(-(sigma_t * dist));
        const Real eps = shadow_epsilon(scene);
        Real visibility = 1;
        if (occluded(scene, Ray{p, dir_light, eps, (1 - eps) * distance(pl.position, p)}, cnt)) visibility = 0;
        const Real jacobian = std::fabs(dot(dir_light, pl.normal)) / distance_squared(p, pl.position) * visibility;
        const Spectrum L_s1 = vmul(Le * rho, exp_term) * jacobian;
        const Real L_s1_pdf = scene.light_dist.pmf[light_id] * pdf_point_on_light(scene, light_id, pl, p);
        return vmul(vmul(vdiv(transmittance, trans_pdf), sigma_s), L_s1 / L_s1_pdf);
    }
    // A ray that left the scene reaches this arm when sigma_t.x is 0 at its origin (t is then inf, or nan for u = 0): the reference would
    // read its empty optional here; there is no vertex to take an emission from, so the sample is zero.
    if (!hit) return {0, 0, 0};
    const Spectrum transmittance = vexp(-(sigma_t * t_hit));   // (trans_pdf is the same expression: the ratio is 1, or 0/0 once it underflows)
    Spectrum Le{0, 0, 0};
    if (scene.shapes[vertex.shape_id].area_light_id >= 0) Le = vertex_emission(scene, vertex, -ray.dir);
    return vmul(vdiv(transmittance, transmittance), Le);
}

// vol_path_tracing (vol_path_tracing.h:503-869), the final renderer.  vol_path_tracing_3 / _4 / _5 return its result in their first
// statement (vol_path_tracing.h:880, 1052, 1297), so versions 3 ... 6 (and an unset version, render.cpp:111) are this function.
// Reference quirks kept: a ray that leaves the scene from vacuum returns ZERO, not the radiance gathered so far (:626);
// an emitter reached by phase / BSDF sampling is weighted with a geometry term of the wrong sign, i.e. zero for a
// front-facing hit (:688-695); no pdf > 0 check after BSDF sampling (:836-841).  nee_p_cache is read before it is
// first written (:520, uninitialised in the reference): zero here.
Spectrum vol_path_tracing(const OScene &scene, int x, int y, pcg32_state &rng, int max_depth_override, bool use_override, Counters *cnt, int *status) {
    const LjCamera &cam = scene.d.camera;
    const int w = cam.width, h = cam.height;
    Real jy = next_pcg32_real(rng);   // g++ evaluates the constructor arguments right to left
    Real jx = next_pcg32_real(rng);
    Ray ray = sample_primary(scene, Vector2{(x + jx) / w, (y + jy) / h});
    ray.tnear = shadow_epsilon(scene);   // get_intersection_epsilon == get_shadow_epsilon (scene.h:99-105)
    RayDifferential ray_diff{0, 0};
    if (cnt) cnt->samples++;
    const LjRenderOptions &opt = scene.d.options;
    const int max_depth = use_override ? max_depth_override : opt.max_depth;
    int current_medium = cam.medium_id;
    Spectrum throughput{1, 1, 1}, radiance{0, 0, 0};
    int bounces = 0;
    Real dir_pdf = 0; Vector3 nee_p_cache{0, 0, 0};
    Spectrum multi_trans_pdf{1, 1, 1};
    Real eta_scale = 1;
    for (;;) {
        bool scatter = false;
        PathVertex vertex; const bool hit = intersect(scene, ray, ray_diff, vertex, cnt);
        Real t_hit = hit ? distance(vertex.position, ray.org) : std::numeric_limits<Real>::infinity();
        Spectrum transmittance{1, 1, 1}, trans_dir_pdf{1, 1, 1}, trans_nee_pdf{1, 1, 1};
        if (current_medium != -1) {
            const LjMedium &med = scene.d.media[current_medium];
            Vector3 majorant = get_majorant(med, ray);
            Real u = next_pcg32_real(rng);
            int channel = std::min(std::max(int(u * 3), 0), 2);
            Real accum_t = 0; int iteration = 0;
            for (;;) {
                if (vget(majorant, channel) <= 0) break;
                if (iteration >= opt.max_null_collisions) break;
                Real t = -std::log(1 - next_pcg32_real(rng)) / vget(majorant, channel);
                Real dt = t_hit - accum_t;
                accum_t = std::min(accum_t + t, t_hit);
                if (t < dt) {
                    Vector3 p = ray.org + ray.dir * accum_t;
                    Vector3 sigma_t = get_sigma_s(med, p) + get_sigma_a(med, p);
                    Vector3 real_prob = vdiv(sigma_t, majorant);
                    Vector3 one_minus{1 - real_prob.x, 1 - real_prob.y, 1 - real_prob.z};
                    Vector3 sigma_n = vmul(majorant, one_minus);
                    Vector3 e = vexp(-(majorant * t));
                    Real mx = vmax(majorant);
                    if (next_pcg32_real(rng) < vget(real_prob, channel)) {   // a real particle
                        scatter = true;
                        transmittance = vmul(transmittance, e / mx);
                        trans_dir_pdf = vmul(trans_dir_pdf, vmul(vmul(e, majorant), real_prob) / mx);
                        ray.org = p;
                        break;
                    }
                    transmittance = vmul(transmittance, vmul(e, sigma_n) / mx);
                    trans_dir_pdf = vmul(trans_dir_pdf, vmul(vmul(e, majorant), one_minus) / mx);
                    trans_nee_pdf = vmul(trans_nee_pdf, vmul(e, majorant) / mx);
                } else {
                    Vector3 e = vexp(-(majorant * dt));
                    transmittance = vmul(transmittance, e); trans_dir_pdf = vmul(trans_dir_pdf, e); trans_nee_pdf = vmul(trans_nee_pdf, e);
                    ray.org = vertex.position;
                    break;
                }
                iteration++;
            }
            multi_trans_pdf = vmul(multi_trans_pdf, trans_dir_pdf);
        } else {
            if (hit) ray.org = vertex.position;
            else return {0, 0, 0};
        }
        throughput = vmul(throughput, transmittance / vavg(trans_dir_pdf));
        if (!scatter && hit && scene.shapes[vertex.shape_id].area_light_id >= 0) {
            Spectrum Le = vertex_emission(scene, vertex, -ray.dir);
            if (bounces == 0) { radiance += vmul(throughput, Le); return radiance; }
            const int light_id = scene.shapes[vertex.shape_id].area_light_id;
            PointAndNormal light_point{vertex.position, vertex.geometry_normal};
            Spectrum pdf_nee = (scene.light_dist.pmf[light_id] * pdf_point_on_light(scene, light_id, light_point, nee_p_cache)) * trans_nee_pdf;
            Real jacobian = std::max(-dot(-ray.dir, light_point.normal), Real(0)) / distance_squared(nee_p_cache, light_point.position);
            Spectrum pdf_phase = (dir_pdf * jacobian) * multi_trans_pdf;
            Spectrum p2 = vmul(pdf_phase, pdf_phase), n2 = vmul(pdf_nee, pdf_nee);
            Spectrum wgt = vdiv(p2, p2 + n2);
            radiance += vmul(vmul(throughput, Le), wgt);
        }
        if (!scatter && hit && vertex.material_id == -1) {   // index-matched surface: change medium, go on
            current_medium = update_medium(scene, vertex, ray, current_medium);
            ray.org = vertex.position;
            bounces++;
            continue;
        }
        if (bounces >= max_depth - 1 && max_depth != -1) break;
        if (cnt) cnt->bounces++;
        if (scatter && current_medium != -1) {
            const LjMedium &med = scene.d.media[current_medium];
            Vector3 sigma_s = get_sigma_s(med, ray.org);
            Spectrum nee = vol_nee(scene, rng, ray.org, current_medium, bounces, -ray.dir, false, vertex, max_depth, cnt, status);
            radiance += vmul(vmul(throughput, sigma_s), nee);
            if (vmax(nee) > 0) nee_p_cache = ray.org;
            Vector2 phase_uv; phase_uv.x = next_pcg32_real(rng); phase_uv.y = next_pcg32_real(rng);
            Vector3 next_dir = phase_sample(med, -ray.dir, phase_uv);
            Real phase_pdf = phase_eval(med, -ray.dir, next_dir);
            throughput = vmul(throughput, (phase_pdf / phase_pdf) * sigma_s);   // eval / pdf, as written (:793-795)
            ray.dir = next_dir;
            dir_pdf = phase_pdf;
            multi_trans_pdf = {1, 1, 1};
        } else if (hit) {
            Spectrum nee = vol_nee(scene, rng, ray.org, current_medium, bounces, -ray.dir, true, vertex, max_depth, cnt, status);
            radiance += vmul(throughput, nee);
            if (vmax(nee) > 0) nee_p_cache = ray.org;
            const LjMaterial &mat = scene.materials[vertex.material_id];
            Vector3 dir_view = -ray.dir;
            Vector2 bsdf_uv; bsdf_uv.x = next_pcg32_real(rng); bsdf_uv.y = next_pcg32_real(rng);
            Real bsdf_w = next_pcg32_real(rng);
            BSDFSampleRecord rec; bool ok;
            if (!bsdf_sample(&scene, mat, dir_view, vertex, bsdf_uv, bsdf_w, ok, rec)) { *status = 1; return {0, 0, 0}; }
            if (!ok) break;
            ray.dir = rec.dir_out;
            if (rec.eta == 0) ray_diff.spread = rd_reflect(ray_diff, vertex.mean_curvature, rec.roughness);
            else {
                ray_diff.spread = rd_refract(ray_diff, vertex.mean_curvature, rec.eta, rec.roughness);
                eta_scale /= (rec.eta * rec.eta);
                current_medium = update_medium(scene, vertex, ray, current_medium);
            }
            Spectrum f; if (!bsdf_eval(&scene, mat, dir_view, rec.dir_out, vertex, f)) { *status = 1; return {0, 0, 0}; }
            Real pdf_bsdf; if (!bsdf_pdf(&scene, mat, dir_view, rec.dir_out, vertex, pdf_bsdf)) { *status = 1; return {0, 0, 0}; }
            throughput = vmul(throughput, f / pdf_bsdf);
        }
        if (bounces >= opt.rr_depth) {
            Real rr_prob = ref_min(ref_vmax((1 / eta_scale) * throughput), Real(0.95));
            if (next_pcg32_real(rng) > rr_prob) break;
            throughput = throughput / rr_prob;
        }
        bounces++;
    }
    return radiance;
}

// render.cpp:12-69 aux_render: one primary ray through the pixel centre, no RNG
static Vector3 aux_pixel(const OScene &s, int x, int y, Counters *cnt) {
    const int w = s.d.camera.width, h = s.d.camera.height;
    Ray ray = sample_primary(s, Vector2{(x + Real(0.5)) / w, (y + Real(0.5)) / h});
    RayDifferential ray_diff{Real(0), Real(0.25) / std::max(w, h)};
    PathVertex vertex;
    if (!intersect(s, ray, ray_diff, vertex, cnt)) return {0, 0, 0};
    switch (s.d.options.integrator) {
        case LJ_INTEGRATOR_DEPTH: { Real d = distance(vertex.position, ray.org); return {d, d, d}; }
        case LJ_INTEGRATOR_SHADING_NORMAL: return vertex.shading_frame.n;
        case LJ_INTEGRATOR_MEAN_CURVATURE: return {vertex.mean_curvature, vertex.mean_curvature, vertex.mean_curvature};
        case LJ_INTEGRATOR_RAY_DIFFERENTIAL: return {ray_diff.radius, ray_diff.spread, Real(0)};
        default: {  // MipmapLevel: only when get_texture(material) (materials/*.inl get_texture_op) is an image texture
            const LjMaterial &m = s.d.materials[vertex.material_id];
            if (m.kind == LJ_MAT_DISNEYCLEARCOAT) return {0, 0, 0};  // disney_clearcoat.inl:108-110: a constant texture
            const LjTexture &t = m.tex[0];
            if (t.kind != LJ_TEX_IMAGE) return {0, 0, 0};
            const Mip &img = s.mips3[t.texture_id];
            Real scaled_footprint = std::max(img.w[0], img.h[0]) * std::max(t.uscale, t.vscale) * vertex.uv_screen_size;
            Real level = std::log2(std::max(scaled_footprint, Real(1e-8f)));
            return {level, level, level};
        }
    }
}

extern "C" {

void *oracle_scene_create(const LjSceneDesc *d) { return scene_create(d); }
void oracle_scene_free(void *s) { delete (OScene *)s; }
void oracle_scene_use_bvh(void *sv, int on) { OScene *s = (OScene *)sv; if (on && s->nodes.empty()) build_bvh(*s); s->use_bvh = on != 0; }

void oracle_scene_tables(void *sv, double *bounds_radius, double *bounds_center, double *shadow_eps, double *light_pmf, double *light_cdf, double *light_power_out) {
    OScene *s = (OScene *)sv;
    *bounds_radius = s->bounds_radius; bounds_center[0] = s->bounds_center.x; bounds_center[1] = s->bounds_center.y; bounds_center[2] = s->bounds_center.z;
    *shadow_eps = shadow_epsilon(*s);
    for (size_t i = 0; i < s->light_dist.pmf.size(); i++) { light_pmf[i] = s->light_dist.pmf[i]; light_power_out[i] = light_power(*s, (int)i); }
    for (size_t i = 0; i < s->light_dist.cdf.size(); i++) light_cdf[i] = s->light_dist.cdf[i];
}
double oracle_mesh_total_area(void *sv, int shape_id) { return ((OScene *)sv)->meshes[shape_id].total_area; }
int oracle_mesh_tri_cdf(void *sv, int shape_id, double *pmf, double *cdf) {
    const TableDist1D &t = ((OScene *)sv)->meshes[shape_id].tri_sampler;
    for (size_t i = 0; i < t.pmf.size(); i++) pmf[i] = t.pmf[i];
    for (size_t i = 0; i < t.cdf.size(); i++) cdf[i] = t.cdf[i];
    return (int)t.pmf.size();
}
int oracle_envmap_dist(void *sv, int light_id, int row, double *total, double *pdf_marg, double *cdf_marg, double *cdf_row, double *pdf_row) {
    const TableDist2D &d = ((OScene *)sv)->env_dist[light_id];
    *total = d.total_values;
    for (int y = 0; y < d.height; y++) pdf_marg[y] = d.pdf_marginals[y];
    for (int y = 0; y <= d.height; y++) cdf_marg[y] = d.cdf_marginals[y];
    for (int x = 0; x <= d.width; x++) cdf_row[x] = d.cdf_rows[(size_t)row * (d.width + 1) + x];
    for (int x = 0; x < d.width; x++) pdf_row[x] = d.pdf_rows[(size_t)row * d.width + x];
    return d.width;
}
int oracle_mip_info(void *sv, int id, int *dims /*2*levels*/, double *level_sums /*3*levels*/) {
    const Mip &m = ((OScene *)sv)->mips3[id];
    for (int l = 0; l < m.levels; l++) {
        dims[2 * l] = m.w[l]; dims[2 * l + 1] = m.h[l];
        Vector3 sum{0, 0, 0}; for (auto &p : m.data[l]) sum += p;
        level_sums[3 * l] = sum.x; level_sums[3 * l + 1] = sum.y; level_sums[3 * l + 2] = sum.z;
    }
    return m.levels;
}

void oracle_pcg32(uint64_t stream, uint64_t seed, int n_u32, uint32_t *u32, int n_f64, double *f64, int n_f32, float *f32, uint64_t *state0, uint64_t *inc) {
    pcg32_state s = init_pcg32(stream, seed ? seed : 0x853c49e6748fea9bULL);
    *state0 = s.state; *inc = s.inc;
    for (int i = 0; i < n_u32; i++) u32[i] = next_pcg32(s);
    for (int i = 0; i < n_f64; i++) f64[i] = next_pcg32_real(s);
    for (int i = 0; i < n_f32; i++) f32[i] = next_pcg32_float(s);
}
void oracle_filter_sample(int kind, double param, const double *rnd, double *out) { Vector2 o = filter_sample(kind, param, Vector2{rnd[0], rnd[1]}); out[0] = o.x; out[1] = o.y; }
void oracle_frame(const double *n, const double *v, double *x, double *y, double *tl, double *tw) {
    Frame f = make_frame(Vector3{n[0], n[1], n[2]}); Vector3 vv{v[0], v[1], v[2]};
    Vector3 a = to_local(f, vv), b = to_world(f, vv);
    for (int i = 0; i < 3; i++) { x[i] = f.x[i]; y[i] = f.y[i]; tl[i] = a[i]; tw[i] = b[i]; }
}
void oracle_raydiff(double radius, double spread, double dist, double curv, double rough, double eta, double *out3) {
    RayDifferential rd{radius, spread};
    out3[0] = rd_transfer(rd, dist); out3[1] = rd_reflect(rd, curv, rough); out3[2] = rd_refract(rd, curv, eta, rough);
}
int oracle_table1d(int n, const double *f, double *pmf, double *cdf, int n_u, const double *u, int *ids) {
    TableDist1D t = make_table_dist_1d(std::vector<Real>(f, f + n));
    for (int i = 0; i < n; i++) pmf[i] = t.pmf[i];
    for (int i = 0; i <= n; i++) cdf[i] = t.cdf[i];
    for (int i = 0; i < n_u; i++) ids[i] = sample_1d(t, u[i]);
    return 0;
}
int oracle_table2d(int w, int h, const double *f, double *cdf_rows, double *pdf_rows, double *cdf_marg, double *pdf_marg, double *total,
                   int n, const double *rnd, double *xy, double *pdfs) {
    TableDist2D t = make_table_dist_2d(std::vector<Real>(f, f + (size_t)w * h), w, h);
    memcpy(cdf_rows, t.cdf_rows.data(), sizeof(Real) * t.cdf_rows.size()); memcpy(pdf_rows, t.pdf_rows.data(), sizeof(Real) * t.pdf_rows.size());
    memcpy(cdf_marg, t.cdf_marginals.data(), sizeof(Real) * t.cdf_marginals.size()); memcpy(pdf_marg, t.pdf_marginals.data(), sizeof(Real) * t.pdf_marginals.size());
    *total = t.total_values;
    for (int i = 0; i < n; i++) { Vector2 p = sample_2d(t, Vector2{rnd[2 * i], rnd[2 * i + 1]}); xy[2 * i] = p.x; xy[2 * i + 1] = p.y; pdfs[i] = pdf_2d(t, p); }
    return 0;
}
void oracle_sample_primary(void *sv, int n, const double *screen_pos, double *org, double *dir) {
    OScene *s = (OScene *)sv;
    for (int i = 0; i < n; i++) {
        Ray r = sample_primary(*s, Vector2{screen_pos[2 * i], screen_pos[2 * i + 1]});
        for (int k = 0; k < 3; k++) { org[3 * i + k] = r.org[k]; dir[3 * i + k] = r.dir[k]; }
    }
}
int oracle_sample_light(void *sv, double u) { return sample_1d(((OScene *)sv)->light_dist, u); }
void oracle_light_sample(void *sv, int light_id, const double *ref, const double *uv, double w, const double *view_dir_in, double footprint,
                         double *pos, double *nrm, double *pdf, double *emission) {
    OScene *s = (OScene *)sv;
    Vector3 r{ref[0], ref[1], ref[2]};
    PointAndNormal pn = sample_point_on_light(*s, light_id, r, Vector2{uv[0], uv[1]}, w);
    *pdf = pdf_point_on_light(*s, light_id, pn, r);
    Spectrum L = light_emission(*s, light_id, Vector3{view_dir_in[0], view_dir_in[1], view_dir_in[2]}, footprint, pn);
    for (int k = 0; k < 3; k++) { pos[k] = pn.position[k]; nrm[k] = pn.normal[k]; emission[k] = L[k]; }
}
// vertex record layout (doubles): position3 gnormal3 fx3 fy3 fn3 st2 uv2 uv_screen_size mean_curvature ray_radius  = 22
static void pack_vertex(const PathVertex &v, double *o) {
    int k = 0;
    for (int i = 0; i < 3; i++) o[k++] = v.position[i];
    for (int i = 0; i < 3; i++) o[k++] = v.geometry_normal[i];
    for (int i = 0; i < 3; i++) o[k++] = v.shading_frame.x[i];
    for (int i = 0; i < 3; i++) o[k++] = v.shading_frame.y[i];
    for (int i = 0; i < 3; i++) o[k++] = v.shading_frame.n[i];
    o[k++] = v.st.x; o[k++] = v.st.y; o[k++] = v.uv.x; o[k++] = v.uv.y;
    o[k++] = v.uv_screen_size; o[k++] = v.mean_curvature; o[k++] = v.ray_radius;
}
static PathVertex unpack_vertex(const double *o) {
    PathVertex v; int k = 0;
    for (int i = 0; i < 3; i++) v.position[i] = o[k++];
    for (int i = 0; i < 3; i++) v.geometry_normal[i] = o[k++];
    for (int i = 0; i < 3; i++) v.shading_frame.x[i] = o[k++];
    for (int i = 0; i < 3; i++) v.shading_frame.y[i] = o[k++];
    for (int i = 0; i < 3; i++) v.shading_frame.n[i] = o[k++];
    v.st.x = o[k++]; v.st.y = o[k++]; v.uv.x = o[k++]; v.uv.y = o[k++];
    v.uv_screen_size = o[k++]; v.mean_curvature = o[k++]; v.ray_radius = o[k++];
    return v;
}
void oracle_make_vertex(void *sv, const double *org, const double *dir, double rd_radius, double rd_spread, int shape_id, int prim_id,
                        float t, float u, float v, const double *Ng, double *vertex22, int *material_id, double *emission3) {
    OScene *s = (OScene *)sv;
    Ray ray{Vector3{org[0], org[1], org[2]}, Vector3{dir[0], dir[1], dir[2]}, 0, std::numeric_limits<Real>::infinity()};
    PathVertex vx = make_vertex(*s, ray, RayDifferential{rd_radius, rd_spread}, shape_id, prim_id, t, u, v, Vector3{Ng[0], Ng[1], Ng[2]});
    pack_vertex(vx, vertex22); *material_id = vx.material_id;
    if (emission3 && s->shapes[shape_id].area_light_id >= 0) { Spectrum L = vertex_emission(*s, vx, -ray.dir); for (int k = 0; k < 3; k++) emission3[k] = L[k]; }
}
// returns 0 ok, 1 material kind not restated yet
int oracle_bsdf(void *sv, const LjMaterial *m, const double *vertex22, const double *dir_in, const double *dir_out, const double *rnd_uv, double rnd_w, int to_view,
                double *eval3, double *pdf, int *sample_valid, double *sample_dir3, double *sample_eta, double *sample_roughness) {
    OScene *s = (OScene *)sv;
    PathVertex vx = unpack_vertex(vertex22);
    Vector3 di{dir_in[0], dir_in[1], dir_in[2]}, dout{dir_out[0], dir_out[1], dir_out[2]};
    Spectrum f; Real p; bool valid; BSDFSampleRecord rec{};
    if (!bsdf_eval(s, *m, di, dout, vx, f, to_view != 0)) return 1;
    bsdf_pdf(s, *m, di, dout, vx, p);
    bsdf_sample(s, *m, di, vx, Vector2{rnd_uv[0], rnd_uv[1]}, rnd_w, valid, rec);
    for (int k = 0; k < 3; k++) { eval3[k] = f[k]; sample_dir3[k] = valid ? rec.dir_out[k] : 0; }
    *pdf = p; *sample_valid = valid; *sample_eta = valid ? rec.eta : 0; *sample_roughness = valid ? rec.roughness : 0;
    return 0;
}

void oracle_intersect(void *sv, int64_t n, const LjRay *rays, LjHit *hits) {
    OScene *s = (OScene *)sv;
    for (int64_t i = 0; i < n; i++) {
        Hit h = scene_intersect(*s, rays[i].org, rays[i].dir, rays[i].tnear, rays[i].tfar);
        hits[i] = LjHit{h.shape_id >= 0 ? h.t : 0.0f, h.u, h.v, h.shape_id, h.prim_id};
    }
}
void oracle_occluded(void *sv, int64_t n, const LjRay *rays, uint8_t *occ) {
    OScene *s = (OScene *)sv;
    for (int64_t i = 0; i < n; i++) occ[i] = scene_occluded(*s, rays[i].org, rays[i].dir, rays[i].tnear, rays[i].tfar);
}
// the reference's one intersection fixture goes through intersect(): (src/tests/intersection.cpp:28-37)
int oracle_intersect_vertex(void *sv, const double *org, const double *dir, double tnear, double tfar, double *vertex22, int *shape_id, int *prim_id) {
    OScene *s = (OScene *)sv;
    Ray ray{Vector3{org[0], org[1], org[2]}, Vector3{dir[0], dir[1], dir[2]}, tnear, tfar};
    PathVertex vx;
    if (!intersect(*s, ray, RayDifferential{}, vx, nullptr)) return 0;
    pack_vertex(vx, vertex22); *shape_id = vx.shape_id; *prim_id = vx.primitive_id;
    return 1;
}

// rng_mode: 0 = one stream per (pixel, sample): stream = (y*w+x)*spp + s  (the GPU's mode)
//           1 = one stream per 16x16 tile consumed sequentially (render.cpp:82-96, the reference's mode)
typedef struct OracleRenderArgs {
    int32_t spp, rng_mode, n_threads, use_max_depth, max_depth;
    int32_t rank, world_size;
    int32_t crop_x0, crop_y0, crop_x1, crop_y1;
    uint64_t seed;
} OracleRenderArgs;
typedef struct OracleStats { uint64_t samples, bounces, rays_closest, rays_shadow; double seconds; int32_t status; } OracleStats;

// rgb: w*h*3 doubles (radiance / spp, render.cpp:94); per_sample (optional): crop_w*crop_h*spp*3 doubles, sample mode only
// media known-answer hooks
void oracle_phase(int phase_kind, double g, const double *dir_in, const double *dir_out, const double *uv, double *eval_out, double *sample3) {
    LjMedium m{}; m.phase_kind = phase_kind; m.g = g;
    *eval_out = phase_eval(m, v3of(dir_in), v3of(dir_out));
    Vector3 smp = phase_sample(m, v3of(dir_in), Vector2{uv[0], uv[1]});
    sample3[0] = smp.x; sample3[1] = smp.y; sample3[2] = smp.z;
}
void oracle_medium_point(void *sv, int medium_id, const double *p, double *sigma_s3, double *sigma_a3) {
    const LjMedium &m = ((OScene *)sv)->d.media[medium_id];
    Vector3 a = get_sigma_s(m, v3of(p)), b = get_sigma_a(m, v3of(p));
    sigma_s3[0] = a.x; sigma_s3[1] = a.y; sigma_s3[2] = a.z; sigma_a3[0] = b.x; sigma_a3[1] = b.y; sigma_a3[2] = b.z;
}
void oracle_medium_majorant(void *sv, int medium_id, const double *org, const double *dir, double tfar, double *out3) {
    const LjMedium &m = ((OScene *)sv)->d.media[medium_id];
    Vector3 r = get_majorant(m, Ray{v3of(org), v3of(dir), Real(0), tfar});
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

int oracle_render(void *sv, const OracleRenderArgs *a, double *rgb, double *per_sample, OracleStats *stats) {
    OScene *s = (OScene *)sv;
    const int w = s->d.camera.width, h = s->d.camera.height;
    const int spp = a->spp > 0 ? a->spp : s->d.options.samples_per_pixel;
    const uint64_t seed = a->seed ? a->seed : 0x853c49e6748fea9bULL;
    const int tile = 16, ntx = (w + tile - 1) / tile, nty = (h + tile - 1) / tile;  // render.cpp:75-77
    bool crop = a->crop_x1 > a->crop_x0 && a->crop_y1 > a->crop_y0;
    int cx0 = crop ? a->crop_x0 : 0, cy0 = crop ? a->crop_y0 : 0, cx1 = crop ? a->crop_x1 : w, cy1 = crop ? a->crop_y1 : h;
    int world = a->world_size > 0 ? a->world_size : 1;
    for (size_t i = 0; i < (size_t)w * h * 3; i++) rgb[i] = 0;
    std::atomic<int> next_tile{0}; std::atomic<int> status{0};
    int nthreads = a->n_threads > 0 ? a->n_threads : (int)std::thread::hardware_concurrency();
    std::vector<Counters> cnts(nthreads);
    auto t_begin = std::chrono::steady_clock::now();
    // parallel_for over tiles: dynamic self-scheduling, one tile per grab (parallel.cpp:183-237)
    auto worker = [&](int tid) {
        Counters &cnt = cnts[tid];
        for (;;) {
            int t = next_tile.fetch_add(1);
            if (t >= ntx * nty) break;
            if (t % world != a->rank) continue;
            int tx = t % ntx, ty = t / ntx;
            pcg32_state rng = init_pcg32((uint64_t)(ty * ntx + tx), seed);  // render.cpp:82
            int x0 = tx * tile, x1 = std::min(x0 + tile, w), y0 = ty * tile, y1 = std::min(y0 + tile, h);
            for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) {
                bool inside = x >= cx0 && x < cx1 && y >= cy0 && y < cy1;
                if (!inside && a->rng_mode == 0) continue;
                Spectrum radiance{0, 0, 0};
                if (s->d.options.integrator < LJ_INTEGRATOR_PATH) {  // aux buffers: one deterministic value per pixel
                    if (!inside) continue;
                    Vector3 c = aux_pixel(*s, x, y, &cnt);
                    size_t o = ((size_t)y * w + x) * 3; rgb[o] = c.x; rgb[o + 1] = c.y; rgb[o + 2] = c.z;
                    continue;
                }
                for (int sidx = 0; sidx < spp; sidx++) {
                    int st = 0;
                    if (a->rng_mode == 0) rng = init_pcg32(((uint64_t)y * w + x) * (uint64_t)spp + sidx, seed);
                    Spectrum L;
                    if (s->d.options.integrator == LJ_INTEGRATOR_VOLPATH) {   // vol_path_render, render.cpp:111-141
                        const int version = s->d.options.vol_path_version;
                        L = version == 1 ? vol_path_tracing_1(*s, x, y, rng, &cnt) : version == 2 ? vol_path_tracing_2(*s, x, y, rng, &cnt)
                                         : vol_path_tracing(*s, x, y, rng, a->max_depth, a->use_max_depth != 0, &cnt, &st);
                        if (!(std::isfinite(L.x) && std::isfinite(L.y) && std::isfinite(L.z))) L = {0, 0, 0};   // render.cpp:138-141: non-finite samples are left out
                    } else L = path_tracing(*s, x, y, rng, a->max_depth, a->use_max_depth != 0, &cnt, &st);
                    if (st) status.store(st);
                    radiance += L;
                    if (per_sample && inside) {
                        size_t o = (((size_t)(y - cy0) * (cx1 - cx0) + (x - cx0)) * spp + sidx) * 3;
                        per_sample[o] = L.x; per_sample[o + 1] = L.y; per_sample[o + 2] = L.z;
                    }
                }
                if (inside) { Spectrum px = radiance / Real(spp); size_t o = ((size_t)y * w + x) * 3; rgb[o] = px.x; rgb[o + 1] = px.y; rgb[o + 2] = px.z; }
            }
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nthreads; i++) th.emplace_back(worker, i);
    worker(0);
    for (auto &t : th) t.join();
    if (stats) {
        *stats = OracleStats{};
        for (auto &c : cnts) { stats->samples += c.samples; stats->bounces += c.bounces; stats->rays_closest += c.rays_closest; stats->rays_shadow += c.rays_shadow; }
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        stats->status = status.load();
    }
    return status.load();
}

} // extern "C"
