// Host-side double-precision math for the front end (scene parsing / flattening).
// Semantics follow the reference where they change bits:
//   * vector / scalar multiplies by the reciprocal          (vector.h:194-197)
//   * normalize(0) == 0                                     (vector.h:249-257)
//   * row-major Matrix4x4, m(i,j)                           (matrix.h:60-66)
#pragma once
#include <cmath>
#include <cstring>

namespace lj {

constexpr double kPi = 3.14159265358979323846;

struct V2 { double x = 0, y = 0; };
struct V3 {
    double x = 0, y = 0, z = 0;
    double &operator[](int i) { return (&x)[i]; }
    const double &operator[](int i) const { return (&x)[i]; }
};
inline V3 operator+(const V3 &a, const V3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3 &a, const V3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(const V3 &a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(const V3 &a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, const V3 &a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(const V3 &a, double s) { double inv = 1.0 / s; return {a.x * inv, a.y * inv, a.z * inv}; }
inline double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3 &a, const V3 &b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double length(const V3 &a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(const V3 &a) { double l = length(a); return l <= 0 ? V3{0, 0, 0} : a / l; }
inline double radians(double deg) { return (kPi / 180.0) * deg; }
inline double degrees(double rad) { return (180.0 / kPi) * rad; }

struct M4 {
    double m[4][4];
    double &operator()(int i, int j) { return m[i][j]; }
    const double &operator()(int i, int j) const { return m[i][j]; }
    static M4 identity() { M4 r; std::memset(&r, 0, sizeof r); for (int i = 0; i < 4; i++) r.m[i][i] = 1; return r; }
    static M4 zero() { M4 r; std::memset(&r, 0, sizeof r); return r; }
};

inline M4 operator*(const M4 &a, const M4 &b) {
    M4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += a(i, k) * b(k, j);
        r(i, j) = s;
    }
    return r;
}

// Adjugate / determinant inverse; returns the zero matrix for a singular input (matrix.h:204-206).
inline M4 inverse(const M4 &a) {
    auto minor3 = [&](int r0, int r1, int r2, int c0, int c1, int c2) {
        return a(r0, c0) * (a(r1, c1) * a(r2, c2) - a(r1, c2) * a(r2, c1)) -
               a(r0, c1) * (a(r1, c0) * a(r2, c2) - a(r1, c2) * a(r2, c0)) +
               a(r0, c2) * (a(r1, c0) * a(r2, c1) - a(r1, c1) * a(r2, c0));
    };
    M4 adj;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        int r[3], c[3], ri = 0, ci = 0;
        for (int k = 0; k < 4; k++) { if (k != i) r[ri++] = k; if (k != j) c[ci++] = k; }
        double mn = minor3(r[0], r[1], r[2], c[0], c[1], c[2]);
        adj(j, i) = ((i + j) & 1) ? -mn : mn;
    }
    double det = a(0, 0) * adj(0, 0) + a(0, 1) * adj(1, 0) + a(0, 2) * adj(2, 0) + a(0, 3) * adj(3, 0);
    if (det == 0) return M4::zero();
    double inv_det = 1.0 / det;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) adj(i, j) *= inv_det;
    return adj;
}

// transform.cpp:5-78
inline M4 translate(const V3 &d) { M4 r = M4::identity(); r(0, 3) = d.x; r(1, 3) = d.y; r(2, 3) = d.z; return r; }
inline M4 scale(const V3 &s) { M4 r = M4::identity(); r(0, 0) = s.x; r(1, 1) = s.y; r(2, 2) = s.z; return r; }
inline M4 rotate(double angle_deg, const V3 &axis) {
    V3 a = normalize(axis);
    double s = std::sin(radians(angle_deg)), c = std::cos(radians(angle_deg));
    M4 r = M4::identity();
    r(0, 0) = a.x * a.x + (1 - a.x * a.x) * c; r(0, 1) = a.x * a.y * (1 - c) - a.z * s; r(0, 2) = a.x * a.z * (1 - c) + a.y * s;
    r(1, 0) = a.x * a.y * (1 - c) + a.z * s; r(1, 1) = a.y * a.y + (1 - a.y * a.y) * c; r(1, 2) = a.y * a.z * (1 - c) - a.x * s;
    r(2, 0) = a.x * a.z * (1 - c) - a.y * s; r(2, 1) = a.y * a.z * (1 - c) + a.x * s; r(2, 2) = a.z * a.z + (1 - a.z * a.z) * c;
    return r;
}
inline M4 look_at(const V3 &pos, const V3 &look, const V3 &up) {
    V3 dir = normalize(look - pos);
    V3 left = normalize(cross(normalize(up), dir));
    V3 new_up = cross(dir, left);
    M4 r = M4::identity();
    for (int i = 0; i < 3; i++) { r(i, 0) = left[i]; r(i, 1) = new_up[i]; r(i, 2) = dir[i]; r(i, 3) = pos[i]; }
    return r;
}
inline M4 perspective(double fov_deg) {
    double cot = 1.0 / std::tan(radians(fov_deg / 2.0));
    M4 r = M4::zero();
    r(0, 0) = cot; r(1, 1) = cot; r(2, 2) = 1; r(2, 3) = -1; r(3, 2) = 1;
    return r;
}
// transform.cpp:80-100
inline V3 xform_point(const M4 &m, const V3 &p) {
    double x = m(0, 0) * p.x + m(0, 1) * p.y + m(0, 2) * p.z + m(0, 3);
    double y = m(1, 0) * p.x + m(1, 1) * p.y + m(1, 2) * p.z + m(1, 3);
    double z = m(2, 0) * p.x + m(2, 1) * p.y + m(2, 2) * p.z + m(2, 3);
    double w = m(3, 0) * p.x + m(3, 1) * p.y + m(3, 2) * p.z + m(3, 3);
    double inv_w = 1.0 / w;
    return {x * inv_w, y * inv_w, z * inv_w};
}
inline V3 xform_vector(const M4 &m, const V3 &v) {
    return {m(0, 0) * v.x + m(0, 1) * v.y + m(0, 2) * v.z,
            m(1, 0) * v.x + m(1, 1) * v.y + m(1, 2) * v.z,
            m(2, 0) * v.x + m(2, 1) * v.y + m(2, 2) * v.z};
}
inline V3 xform_normal(const M4 &inv, const V3 &n) {
    return {inv(0, 0) * n.x + inv(1, 0) * n.y + inv(2, 0) * n.z,
            inv(0, 1) * n.x + inv(1, 1) * n.y + inv(2, 1) * n.z,
            inv(0, 2) * n.x + inv(1, 2) * n.y + inv(2, 2) * n.z};
}

} // namespace lj
