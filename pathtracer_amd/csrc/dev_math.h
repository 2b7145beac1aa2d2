/*
 * dev_math.h — device-side (gfx950) math of the hot path: the arithmetic of the reference's
 * src/core/geometry.h and src/core/transform.h, expression by expression (operation order and
 * float/double promotion points kept; compiled with -ffp-contract=off), so that everything up
 * to the first libm call is bit-identical to a host build of the reference.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

#define PD __device__ __forceinline__

#define kPi 3.14159265358979323846
#define kInvPi 0.31830988618379067154
#define kInv2Pi 0.15915494309189533577
#define kInv4Pi 0.07957747154594766788
#define kPiOver2 1.57079632679489661923
#define kPiOver4 0.78539816339744830961
#define kOneMinusEpsilon 0.99999994f
#define kShadowEpsilon 0.0001f
#define kMachineEpsilon 5.96046e-08f
#define kInfinity FLT_MAX

struct f3 {
    float x, y, z;
    PD f3() : x(0), y(0), z(0) {}
    PD explicit f3(float a) : x(a), y(a), z(a) {}
    PD f3(float a, float b, float c) : x(a), y(b), z(c) {}
};
struct f2 {
    float x, y;
    PD f2() : x(0), y(0) {}
    PD f2(float a, float b) : x(a), y(b) {}
};

PD float fmaxr(float a, float b) { return a < b ? b : a; } /* geometry.h:84 Max */
PD float fminr(float a, float b) { return a < b ? a : b; } /* geometry.h:85 Min */
PD float absr(float v) { return v > 0 ? v : -v; }          /* geometry.h:86 Absf */
/* IsZero / IsUnsafeHit compare Absf(a) with a double literal (1e-9, 1e-6).  The floats below such
 * a literal are exactly the floats below the first float >= it, so the conversion can be skipped. */
PD bool is_zero(float a) { return absr(a) < 1.00000008e-09f; }
PD bool is_unsafe_hit(float a) { return absr(a) < 1.00000011e-06f; }
static_assert((double)1.00000008e-09f >= 1e-9 && (double)9.99999972e-10f < 1e-9, "IsZero threshold");
static_assert((double)1.00000011e-06f >= 1e-6 && (double)9.99999997e-07f < 1e-6, "IsUnsafeHit threshold");
static_assert(9.99999972e-10f < 1.00000008e-09f && 9.99999997e-07f < 1.00000011e-06f, "adjacent floats");
PD float clampr(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
PD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
PD float gammar(int n) { return ((float)n * kMachineEpsilon) / (1 - (float)n * kMachineEpsilon); }

PD float comp(const f3 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
PD void set_comp(f3 &v, int i, float s) { if (i == 0) v.x = s; else if (i == 1) v.y = s; else v.z = s; }

PD f3 operator+(const f3 &a, const f3 &b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
PD f3 operator-(const f3 &a, const f3 &b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
PD f3 operator-(const f3 &a) { return f3(-a.x, -a.y, -a.z); }
PD f3 operator*(const f3 &a, const f3 &b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
PD f3 operator*(const f3 &a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
PD f3 operator*(float s, const f3 &a) { return f3(a.x * s, a.y * s, a.z * s); }
PD f3 operator/(const f3 &a, float f) { float inv = 1.0f / f; return f3(a.x * inv, a.y * inv, a.z * inv); }
PD f3 divv(const f3 &a, const f3 &v) { float ix = 1.0f / v.x, iy = 1.0f / v.y, iz = 1.0f / v.z; return f3(a.x * ix, a.y * iy, a.z * iz); }
PD f3 &operator+=(f3 &a, const f3 &b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
PD f3 &operator*=(f3 &a, const f3 &b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; return a; }
PD f3 &operator*=(f3 &a, float s) { a.x *= s; a.y *= s; a.z *= s; return a; }
PD f2 operator+(const f2 &a, const f2 &b) { return f2(a.x + b.x, a.y + b.y); }
PD f2 operator-(const f2 &a, const f2 &b) { return f2(a.x - b.x, a.y - b.y); }
PD f2 operator*(const f2 &a, float s) { return f2(a.x * s, a.y * s); }
PD f2 operator*(float s, const f2 &a) { return f2(a.x * s, a.y * s); }

PD float dot(const f3 &a, const f3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PD float absdot(const f3 &a, const f3 &b) { return absr(dot(a, b)); }
PD float length2(const f3 &v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
PD float length(const f3 &v) { return sqrtf(length2(v)); }
PD f3 normalize(const f3 &v) { return v / length(v); }
PD f3 abs3(const f3 &v) { return f3(absr(v.x), absr(v.y), absr(v.z)); }
/* geometry.h:642-648: double products and differences, narrowed per component */
PD f3 cross(const f3 &a, const f3 &b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return f3((float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx)));
}
PD float max_comp(const f3 &v) { return fmaxr(v.x, fmaxr(v.y, v.z)); }
PD int max_dim(const f3 &v) { return (v.x > v.y) ? ((v.x > v.z) ? 0 : 2) : ((v.y > v.z) ? 1 : 2); }
PD bool is_black(const f3 &s) { return is_zero(s.x) && is_zero(s.y) && is_zero(s.z); }
PD float dist2(const f3 &a, const f3 &b) { return length2(a - b); }
PD float dist(const f3 &a, const f3 &b) { return length(a - b); }
PD f3 faceforward(const f3 &n, const f3 &v) { return (dot(n, v) < 0.f) ? -n : n; }
PD f3 exp3(const f3 &v) { return f3(expf(v.x), expf(v.y), expf(v.z)); }
PD f3 sqrt3(const f3 &v) { return f3(sqrtf(v.x), sqrtf(v.y), sqrtf(v.z)); }

PD uint32_t f2bits(float f) { return __float_as_uint(f); }
PD float bits2f(uint32_t u) { return __uint_as_float(u); }
PD float next_up(float v) { /* geometry.h:129-138 */
    if (isinf(v) && v > 0.) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = f2bits(v);
    if (v >= 0) ++ui; else --ui;
    return bits2f(ui);
}
PD float next_down(float v) { /* geometry.h:140-149 */
    if (isinf(v) && v < 0.) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = f2bits(v);
    if (v > 0) --ui; else ++ui;
    return bits2f(ui);
}
PD bool quadratic(float a, float b, float c, float *t0, float *t1) { /* geometry.h:193-206 */
    double discrim = (double)b * (double)b - 4 * (double)a * (double)c;
    if (discrim < 0) return false;
    double rootDiscrim = sqrt(discrim);
    double q;
    if (b < 0) q = -.5 * (b - rootDiscrim);
    else q = -.5 * (b + rootDiscrim);
    *t0 = q / a;
    *t1 = c / q;
    if (*t0 > *t1) { float s = *t0; *t0 = *t1; *t1 = s; }
    return true;
}
PD void coordinate_system(const f3 &v1, f3 *v2, f3 *v3) { /* geometry.h:716-729 */
    if (absr(v1.x) > absr(v1.y)) {
        float f = sqrtf(v1.x * v1.x + v1.z * v1.z);
        *v2 = f3(-v1.z, 0, v1.x) / f;
    } else {
        float f = sqrtf(v1.z * v1.z + v1.y * v1.y);
        *v2 = f3(0, v1.z, -v1.y) / f;
    }
    *v3 = cross(v1, *v2);
}
PD f3 offset_ray_origin(const f3 &p, const f3 &pError, const f3 &n, const f3 &w) { /* geometry.h:1527-1542 */
    float d = dot(abs3(n), pError) + 0.0001;
    f3 offset = d * n;
    if (dot(w, n) < 0) offset = -offset;
    f3 po = p + offset;
    if (offset.x > 0) po.x = next_up(po.x); else if (offset.x < 0) po.x = next_down(po.x);
    if (offset.y > 0) po.y = next_up(po.y); else if (offset.y < 0) po.y = next_down(po.y);
    if (offset.z > 0) po.z = next_up(po.z); else if (offset.z < 0) po.z = next_down(po.z);
    return po;
}
PD float spherical_theta(const f3 &v) { return acosf(clampr(v.z, -1, 1)); }
PD float spherical_phi(const f3 &v) { float p = atan2f(v.y, v.x); return (p < 0) ? (p + 2 * kPi) : p; }
PD f3 spherical_direction(float sinTheta, float cosTheta, float phi) { return f3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta); }
PD f3 spherical_direction(float sinTheta, float cosTheta, float phi, const f3 &x, const f3 &y, const f3 &z) {
    return sinTheta * cosf(phi) * x + sinTheta * sinf(phi) * y + cosTheta * z;
}
PD bool same_hemisphere(const f3 &w, const f3 &wp) { return w.z * wp.z > 0; }
PD f2 concentric_sample_disk(const f2 &u) { /* geometry.h:1566-1581 */
    f2 uOffset = 2.f * u - f2(1, 1);
    if (uOffset.x == 0 && uOffset.y == 0) return f2(0, 0);
    float theta, r;
    if (absr(uOffset.x) > absr(uOffset.y)) {
        r = uOffset.x;
        theta = kPiOver4 * (uOffset.y / uOffset.x);
    } else {
        r = uOffset.y;
        theta = kPiOver2 - kPiOver4 * (uOffset.x / uOffset.y);
    }
    return r * f2(cosf(theta), sinf(theta));
}
PD f3 cosine_sample_hemisphere(const f2 &u) {
    f2 d = concentric_sample_disk(u);
    float z = sqrtf(fmaxr(0.f, 1 - d.x * d.x - d.y * d.y));
    return f3(d.x, d.y, z);
}
PD float power_heuristic(int nf, float fPdf, int ng, float gPdf) {
    float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}
PD f3 sample_sphere(const f2 &u) { /* geometry.h:1599-1604 */
    float usqrt = 2 * sqrtf(u.y * (1 - u.y));
    float utheta = 2 * kPi * u.x;
    return f3(cosf(utheta) * usqrt, sinf(utheta) * usqrt, 1 - 2 * u.y);
}
PD float uniform_cone_pdf(float cosThetaMax) { return 1 / (2 * kPi * (1 - cosThetaMax)); }
PD f2 uniform_sample_triangle(const f2 &u) { float su0 = sqrtf(u.x); return f2(1 - su0, u.y * su0); }

PD float cos_theta(const f3 &w) { return w.z; }
PD float cos2_theta(const f3 &w) { return w.z * w.z; }
PD float abs_cos_theta(const f3 &w) { return absr(w.z); }
PD float sin2_theta(const f3 &w) { return fmaxr(0.f, 1.f - cos2_theta(w)); }
PD float sin_theta(const f3 &w) { return sqrtf(sin2_theta(w)); }
PD float tan_theta(const f3 &w) { return sin_theta(w) / cos_theta(w); }
PD float tan2_theta(const f3 &w) { return sin2_theta(w) / cos2_theta(w); }
PD float cos_phi(const f3 &w) { float s = sin_theta(w); return (s == 0) ? 1 : clampr(w.x / s, -1, 1); }
PD float sin_phi(const f3 &w) { float s = sin_theta(w); return (s == 0) ? 0 : clampr(w.y / s, -1, 1); }
PD float cos2_phi(const f3 &w) { return cos_phi(w) * cos_phi(w); }
PD float sin2_phi(const f3 &w) { return sin_phi(w) * sin_phi(w); }
PD f3 reflect(const f3 &wo, const f3 &n) { return -wo + 2 * dot(wo, n) * n; }
PD bool refract(const f3 &wi, const f3 &n, float eta, f3 *wt) {
    float cosThetaI = dot(n, wi);
    float sin2ThetaI = fmaxr(0.f, float(1 - cosThetaI * cosThetaI));
    float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    float cosThetaT = sqrtf(1 - sin2ThetaT);
    *wt = eta * -wi + (eta * cosThetaI - cosThetaT) * n;
    return true;
}

/* ---- affine transform: rows 0..2 of m and mInv (row 3 is (0,0,0,1), checked at upload, so
 * the reference's wp == 1 branch of transform.h:142 is always the one taken) ------------- */
struct Xf34 {
    float m[12];
    float mi[12];
};
PD f3 xf_point(const float *m, const f3 &p) { /* transform.h:135-148 */
    return f3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
PD f3 xf_vec(const float *m, const f3 &v) { /* transform.h:150-156 */
    return f3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
PD f3 xf_normal(const float *mi, const f3 &n) { /* transform.h:158-164: transposed inverse */
    return f3(mi[0] * n.x + mi[4] * n.y + mi[8] * n.z, mi[1] * n.x + mi[5] * n.y + mi[9] * n.z, mi[2] * n.x + mi[6] * n.y + mi[10] * n.z);
}
PD f3 xf_point_err(const float *m, const f3 &p, f3 *pError) { /* transform.h:192-213 */
    float x = p.x, y = p.y, z = p.z;
    float xp = (m[0] * x + m[1] * y) + (m[2] * z + m[3]);
    float yp = (m[4] * x + m[5] * y) + (m[6] * z + m[7]);
    float zp = (m[8] * x + m[9] * y) + (m[10] * z + m[11]);
    float xa = (absr(m[0] * x) + absr(m[1] * y) + absr(m[2] * z) + absr(m[3]));
    float ya = (absr(m[4] * x) + absr(m[5] * y) + absr(m[6] * z) + absr(m[7]));
    float za = (absr(m[8] * x) + absr(m[9] * y) + absr(m[10] * z) + absr(m[11]));
    *pError = gammar(3) * f3(xa, ya, za);
    return f3(xp, yp, zp);
}
PD f3 xf_point_err2(const float *m, const f3 &pt, const f3 &ptError, f3 *absError) { /* transform.h:215-248 */
    float x = pt.x, y = pt.y, z = pt.z;
    float xp = (m[0] * x + m[1] * y) + (m[2] * z + m[3]);
    float yp = (m[4] * x + m[5] * y) + (m[6] * z + m[7]);
    float zp = (m[8] * x + m[9] * y) + (m[10] * z + m[11]);
    float e[3];
    for (int r = 0; r < 3; r++) {
        const float *q = m + 4 * r;
        e[r] = (gammar(3) + 1.0f) * (absr(q[0]) * ptError.x + absr(q[1]) * ptError.y + absr(q[2]) * ptError.z) +
               gammar(3) * (absr(q[0] * x) + absr(q[1] * y) + absr(q[2] * z) + absr(q[3]));
    }
    *absError = f3(e[0], e[1], e[2]);
    return f3(xp, yp, zp);
}
/* transform.h:295-307: object-space ray; direction error is computed by the reference but never
 * read afterwards, so only the origin error is formed */
PD void xf_ray_err(const float *m, const f3 &ro, const f3 &rd, f3 *o_out, f3 *d_out) {
    f3 oErr;
    f3 o = xf_point_err(m, ro, &oErr);
    f3 d = xf_vec(m, rd);
    float l2 = length2(d);
    if (l2 > 0) {
        float dt = dot(abs3(d), oErr) / l2;
        o += d * dt;
    }
    *o_out = o;
    *d_out = d;
}
