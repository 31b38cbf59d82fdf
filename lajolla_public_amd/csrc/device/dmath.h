// Float vector math, frames and PCG32 for the device path (also compiled for the host by the test twin).
// Semantics mirror the reference where they change results: v / s multiplies by the reciprocal
// (vector.h:194-197), normalize(0) = 0 (vector.h:249-257), Frisvad frame with the -1+1e-6 branch (frame.h:11-22).
#pragma once
#include "dtypes.h"
#include <math.h>

namespace ljd {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr float kInvTwoPi = 0.15915494309189533577f;

struct f3 { float x, y, z; };
LJ_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
LJ_HD f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
LJ_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
LJ_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
LJ_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
LJ_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
LJ_HD f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
LJ_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
// 1 / s and 1 / sqrt(d) for scaling vectors: on the device the one-instruction, 1-ulp v_rcp_f32 / v_rsq_f32 (an IEEE
// division costs eleven instructions, sqrt + division twenty); the host twin keeps the libm forms.
LJ_HD float recip_fast(float s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(s);
#else
    return 1.0f / s;
#endif
}
// a / b correctly rounded whatever division mode the translation unit is compiled in (hipcc can build fp32 division as a 2.5-ulp
// sequence, see build.py): the quotient of two floats computed in double and rounded once more is the IEEE float
// quotient (53 >= 2 * 24 + 2 bits).  Used where a value must match the CPU oracle bit for bit: the hit distance and barycentrics.
LJ_HD float div_ieee(float a, float b) { return (float)((double)a / (double)b); }
LJ_HD f3 operator/(f3 a, float s) { float inv = recip_fast(s); return mk3(a.x * inv, a.y * inv, a.z * inv); }
LJ_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LJ_HD f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
LJ_HD float length(f3 a) { return sqrtf(dot(a, a)); }
LJ_HD f3 normalize(f3 a) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float d = dot(a, a);
    if (!(d > 0.0f)) return mk3(0, 0, 0);
    return a * __builtin_amdgcn_rsqf(d);
#else
    float l = length(a); if (l <= 0.0f) return mk3(0, 0, 0); return a / l;
#endif
}
LJ_HD float max3(f3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
LJ_HD float luminance(f3 s) { return s.x * 0.212671f + s.y * 0.715160f + s.z * 0.072169f; }
LJ_HD float clampf(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }
LJ_HD float modulof(float a, float b) { float r = fmodf(a, b); return (r < 0.0f) ? r + b : r; }
LJ_HD int moduloi(int a, int b) { int r = a % b; return (r < 0) ? r + b : r; }

// sin and cos of 2 pi r for a random number r in [0, 1): the azimuth of every direction sampler.  gfx950's v_sin_f32 / v_cos_f32 take
// their argument in revolutions, so on the device this is one instruction each instead of libm's range reduction + polynomial
// (~40 instructions each; -2.6 % per cbox render, every parity bar unchanged); the host twin keeps sinf / cosf.
LJ_HD void sincos_2pi(float r, float &s, float &c) {
#if defined(__HIP_DEVICE_COMPILE__)
    s = __builtin_amdgcn_sinf(r); c = __builtin_amdgcn_cosf(r);
#else
    const float phi = kTwoPi * r; s = sinf(phi); c = cosf(phi);
#endif
}

struct Frame3 { f3 x, y, n; };
LJ_HD void coordinate_system(f3 n, f3 &a_out, f3 &b_out) {
    if (n.z < -1.0f + 1e-6f) { a_out = mk3(0, -1, 0); b_out = mk3(-1, 0, 0); }
    else {
        float a = 1.0f / (1.0f + n.z), b = -n.x * n.y * a;
        a_out = mk3(1.0f - n.x * n.x * a, b, -n.x);
        b_out = mk3(b, 1.0f - n.y * n.y * a, -n.y);
    }
}
LJ_HD Frame3 make_frame(f3 n) { Frame3 f; f.n = n; coordinate_system(n, f.x, f.y); return f; }
LJ_HD Frame3 flip(Frame3 f) { Frame3 r; r.x = -f.x; r.y = -f.y; r.n = -f.n; return r; }
LJ_HD f3 to_local(const Frame3 &f, f3 v) { return mk3(dot(v, f.x), dot(v, f.y), dot(v, f.n)); }
LJ_HD f3 to_world(const Frame3 &f, f3 v) { return f.x * v.x + f.y * v.y + f.n * v.z; }

// ---- PCG32 (pcg.h:16-41): integer arithmetic, bit-exact with the reference.
LJ_HD uint32_t pcg32_next(uint64_t &state, uint64_t inc) {
    uint64_t old = state;
    state = old * 6364136223846793005ULL + (inc | 1ULL);
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
}
// n / d for a launch-constant d (DFastDiv, dtypes.h)
LJ_HD uint32_t fast_div(uint32_t n, const DFastDiv &f) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __umulhi(f.m, n);
#else
    const uint32_t t = (uint32_t)(((uint64_t)f.m * n) >> 32);
#endif
    return (t + ((n - t) >> f.s1)) >> f.s2;
}
LJ_HD uint64_t pcg32_inc(uint64_t stream_id) { return (stream_id << 1u) | 1u; }
LJ_HD uint64_t pcg32_init(uint64_t stream_id, uint64_t seed) {
    uint64_t inc = pcg32_inc(stream_id), state = 0;
    pcg32_next(state, inc);
    state += seed;
    pcg32_next(state, inc);
    return state;
}
// The reference draws doubles r/2^32 (pcg.h:61-68).  The device uses the same 32 random bits rounded to float;
// a value that would round up to 1.0f is clamped to the largest float below 1 so [0,1) holds.
LJ_HD float pcg32_real(uint64_t &state, uint64_t inc) {
    float u = (float)pcg32_next(state, inc) * 2.3283064365386963e-10f;
    return fminf(u, 0.99999994f);
}

} // namespace ljd
