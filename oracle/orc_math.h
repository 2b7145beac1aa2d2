/*
 * orc_math.h — ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * CPU restatement of the reference's L1 math: src/core/geometry.h, src/core/transform.h,
 * src/core/sampling.h.  Written so that a g++ x86-64 build (SSE2 scalar math, no FMA
 * contraction) performs the same IEEE operations, in the same order and at the same
 * precision (float vs double promotion), as a g++ host build of the reference would.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors and cannot be built in
 * this image without stand-in CUDA headers, so this restatement is checked only against
 * itself, analytic properties, and the HIP path.  Each function cites what it restates.
 */
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

typedef float Float; /* geometry.h:50 */

/* geometry.h:33-42: these are double literals in the reference; expressions that mix
 * them with Float promote to double — kept as double here on purpose. */
#define ORC_Pi 3.14159265358979323846
#define ORC_InvPi 0.31830988618379067154
#define ORC_Inv2Pi 0.15915494309189533577
#define ORC_Inv4Pi 0.07957747154594766788
#define ORC_PiOver2 1.57079632679489661923
#define ORC_PiOver4 0.78539816339744830961
#define ORC_OneMinusEpsilon 0.99999994f
#define ORC_ShadowEpsilon 0.0001f
#define ORC_MachineEpsilon 5.96046e-08f
#define ORC_Infinity FLT_MAX

/* geometry.h:84-93 */
inline Float Max(Float a, Float b) { return a < b ? b : a; }
inline Float Min(Float a, Float b) { return a < b ? a : b; }
inline Float Absf(Float v) { return v > 0 ? v : -v; }
inline bool IsNaN(Float v) { return v != v; }
inline bool IsZero(Float a) { return Absf(a) < 1e-9; }       /* float vs double literal */
inline bool IsUnsafeHit(Float a) { return Absf(a) < 1e-6; }
inline Float Radians(Float deg) { return (ORC_Pi / 180) * deg; } /* double math, narrowed on return */

/* geometry.h:173-178 (all uses on this path are Float with int/float bounds) */
inline Float Clampf(Float v, Float lo, Float hi) {
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}
inline int Clampi(int v, int lo, int hi) {
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

/* geometry.h:180-183: gamma(n) in Float arithmetic */
inline Float gammaf(int n) { return ((Float)n * ORC_MachineEpsilon) / (1 - (Float)n * ORC_MachineEpsilon); }

/* geometry.h:105-149 */
inline uint32_t FloatToBits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float BitsToFloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline float NextFloatUp(float v) {
    if (std::isinf(v) && v > 0.) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = FloatToBits(v);
    if (v >= 0) ++ui; else --ui;
    return BitsToFloat(ui);
}
inline float NextFloatDown(float v) {
    if (std::isinf(v) && v < 0.) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = FloatToBits(v);
    if (v > 0) --ui; else ++ui;
    return BitsToFloat(ui);
}

/* geometry.h:193-206 */
inline bool Quadratic(Float a, Float b, Float c, Float *t0, Float *t1) {
    double discrim = (double)b * (double)b - 4 * (double)a * (double)c;
    if (discrim < 0) return false;
    double rootDiscrim = std::sqrt(discrim);
    double q;
    if (b < 0) q = -.5 * (b - rootDiscrim);
    else q = -.5 * (b + rootDiscrim);
    *t0 = q / a;
    *t1 = c / q;
    if (*t0 > *t1) { Float s = *t0; *t0 = *t1; *t1 = s; }
    return true;
}

/* One struct stands for vec3f / Point3f / Normal3f / Spectrum (geometry.h:295-426,
 * 773-886, 1110-1202); what differs between them in the reference is only how a
 * Transform applies, which the xf_* functions below keep apart. */
struct V3 {
    Float x, y, z;
    V3() : x(0), y(0), z(0) {}
    explicit V3(Float a) : x(a), y(a), z(a) {}
    V3(Float a, Float b, Float c) : x(a), y(b), z(c) {}
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
struct V2 {
    Float x, y;
    V2() : x(0), y(0) {}
    V2(Float a, Float b) : x(a), y(b) {}
    Float operator[](int i) const { return i == 0 ? x : y; }
    Float &operator[](int i) { return i == 0 ? x : y; }
};

inline V3 operator+(const V3 &a, const V3 &b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3 &a, const V3 &b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(const V3 &a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(const V3 &a, const V3 &b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
/* scalar on either side is narrowed to Float first (geometry.h:409-411, 604-607) */
inline V3 operator*(const V3 &a, Float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(Float s, const V3 &a) { return V3(a.x * s, a.y * s, a.z * s); }
/* division by a scalar multiplies by the Float reciprocal (geometry.h:337-344) */
inline V3 operator/(const V3 &a, Float f) { Float inv = (Float)1 / f; return V3(a.x * inv, a.y * inv, a.z * inv); }
/* component-wise division, geometry.h:357-363 */
inline V3 divv(const V3 &a, const V3 &v) {
    Float ix = (Float)1 / v.x, iy = (Float)1 / v.y, iz = (Float)1 / v.z;
    return V3(a.x * ix, a.y * iy, a.z * iz);
}
inline V3 &operator+=(V3 &a, const V3 &b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline V3 &operator*=(V3 &a, const V3 &b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; return a; }
inline V3 &operator*=(V3 &a, Float s) { a.x *= s; a.y *= s; a.z *= s; return a; }

inline V2 operator+(const V2 &a, const V2 &b) { return V2(a.x + b.x, a.y + b.y); }
inline V2 operator-(const V2 &a, const V2 &b) { return V2(a.x - b.x, a.y - b.y); }
inline V2 operator*(const V2 &a, Float s) { return V2(a.x * s, a.y * s); }
inline V2 operator*(Float s, const V2 &a) { return V2(a.x * s, a.y * s); }

inline Float Dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Float AbsDot(const V3 &a, const V3 &b) { return Absf(Dot(a, b)); }
inline Float LengthSquared(const V3 &v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
inline Float Length(const V3 &v) { return sqrtf(LengthSquared(v)); } /* sqrt(Float) -> Float overload */
inline V3 Normalize(const V3 &v) { return v / Length(v); }
inline V3 Abs(const V3 &v) { return V3(Absf(v.x), Absf(v.y), Absf(v.z)); }
/* geometry.h:642-648: products and differences in double, narrowed per component */
inline V3 Cross(const V3 &a, const V3 &b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V3((Float)((ay * bz) - (az * by)), (Float)((az * bx) - (ax * bz)), (Float)((ax * by) - (ay * bx)));
}
inline Float MaxComponent(const V3 &v) { return Max(v.x, Max(v.y, v.z)); }
inline int MaxDimension(const V3 &v) { return (v.x > v.y) ? ((v.x > v.z) ? 0 : 2) : ((v.y > v.z) ? 1 : 2); }
inline V3 Permute(const V3 &v, int x, int y, int z) { return V3(v[x], v[y], v[z]); }
inline bool IsBlack(const V3 &s) { return IsZero(s.x) && IsZero(s.y) && IsZero(s.z); }
inline Float DistanceSquared(const V3 &a, const V3 &b) { return LengthSquared(a - b); }
inline Float Distance(const V3 &a, const V3 &b) { return Length(a - b); }
inline V3 Faceforward(const V3 &n, const V3 &v) { return (Dot(n, v) < 0.f) ? -n : n; }
inline V3 ExpV(const V3 &v) { return V3(expf(v.x), expf(v.y), expf(v.z)); }
inline V3 SqrtV(const V3 &v) { return V3(sqrtf(v.x), sqrtf(v.y), sqrtf(v.z)); }
inline Float SpectrumY(const V3 &s) { return 0.212671f * s.x + 0.715160f * s.y + 0.072169f * s.z; }

/* geometry.h:716-729 */
inline void CoordinateSystem(const V3 &v1, V3 *v2, V3 *v3) {
    if (Absf(v1.x) > Absf(v1.y)) {
        Float f = sqrtf(v1.x * v1.x + v1.z * v1.z);
        *v2 = V3(-v1.z, 0, v1.x) / f;
    } else {
        Float f = sqrtf(v1.z * v1.z + v1.y * v1.y);
        *v2 = V3(0, v1.z, -v1.y) / f;
    }
    *v3 = Cross(v1, *v2);
}

/* geometry.h:1264-1283.  medium: index into the scene's medium table, -1 = none. */
struct Ray {
    V3 o, d;
    mutable Float tMax;
    Float time;
    int medium;
    Ray() : tMax(ORC_Infinity), time(0.f), medium(-1) {}
    Ray(const V3 &o_, const V3 &d_, Float tMax_ = ORC_Infinity, Float time_ = 0.f, int med = -1)
        : o(o_), d(d_), tMax(tMax_), time(time_), medium(med) {}
    V3 at(Float t) const { return o + d * t; }
};

/* geometry.h:1310-1407 */
struct Bounds3 {
    V3 pMin, pMax;
    Bounds3() : pMin(FLT_MAX, FLT_MAX, FLT_MAX), pMax(FLT_MIN, FLT_MIN, FLT_MIN) {} /* sic: FLT_MIN */
    Bounds3(const V3 &p1, const V3 &p2)
        : pMin(Min(p1.x, p2.x), Min(p1.y, p2.y), Min(p1.z, p2.z)),
          pMax(Max(p1.x, p2.x), Max(p1.y, p2.y), Max(p1.z, p2.z)) {}
    static Bounds3 point(const V3 &p) { Bounds3 b; b.pMin = p; b.pMax = p; return b; }
};
inline Bounds3 Union(const Bounds3 &b, const V3 &p) {
    Bounds3 r;
    r.pMin = V3(Min(b.pMin.x, p.x), Min(b.pMin.y, p.y), Min(b.pMin.z, p.z));
    r.pMax = V3(Max(b.pMax.x, p.x), Max(b.pMax.y, p.y), Max(b.pMax.z, p.z));
    return r;
}
inline Bounds3 Union(const Bounds3 &a, const Bounds3 &b) {
    Bounds3 r;
    r.pMin = V3(Min(a.pMin.x, b.pMin.x), Min(a.pMin.y, b.pMin.y), Min(a.pMin.z, b.pMin.z));
    r.pMax = V3(Max(a.pMax.x, b.pMax.x), Max(a.pMax.y, b.pMax.y), Max(a.pMax.z, b.pMax.z));
    return r;
}
inline bool Inside(const V3 &p, const Bounds3 &b) {
    return (p.x >= b.pMin.x && p.x <= b.pMax.x && p.y >= b.pMin.y && p.y <= b.pMax.y && p.z >= b.pMin.z &&
            p.z <= b.pMax.z);
}
/* geometry.h:1481-1499 — slab test with 3 divides and tFar *= 1+2*gamma(3) */
inline bool BoundsIntersectP(const Bounds3 &b, const Ray &ray, Float *hitt0 = nullptr, Float *hitt1 = nullptr) {
    Float t0 = 0, t1 = ray.tMax;
    for (int i = 0; i < 3; ++i) {
        Float invRayDir = 1 / ray.d[i];
        Float tNear = (b.pMin[i] - ray.o[i]) * invRayDir;
        Float tFar = (b.pMax[i] - ray.o[i]) * invRayDir;
        if (tNear > tFar) { Float s = tNear; tNear = tFar; tFar = s; }
        tFar *= 1 + 2 * gammaf(3);
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return false;
    }
    if (hitt0) *hitt0 = t0;
    if (hitt1) *hitt1 = t1;
    return true;
}
/* geometry.h:1388-1391 */
inline void BoundingSphere(const Bounds3 &b, V3 *center, Float *radius) {
    *center = (b.pMin + b.pMax) / 2;
    *radius = Inside(*center, b) ? Distance(*center, b.pMax) : 0;
}

/* geometry.h:1527-1542 */
inline V3 OffsetRayOrigin(const V3 &p, const V3 &pError, const V3 &n, const V3 &w) {
    Float d = Dot(Abs(n), pError) + 0.0001; /* double add, narrowed */
    V3 offset = d * n;
    if (Dot(w, n) < 0) offset = -offset;
    V3 po = p + offset;
    for (int i = 0; i < 3; ++i) {
        if (offset[i] > 0) po[i] = NextFloatUp(po[i]);
        else if (offset[i] < 0) po[i] = NextFloatDown(po[i]);
    }
    return po;
}

/* geometry.h:1545-1554 */
inline Float SphericalTheta(const V3 &v) { return acosf(Clampf(v.z, -1, 1)); }
inline Float SphericalPhi(const V3 &v) {
    Float p = atan2f(v.y, v.x);
    return (p < 0) ? (p + 2 * ORC_Pi) : p;
}
/* geometry.h:1469-1479 */
inline V3 SphericalDirection(Float sinTheta, Float cosTheta, Float phi) {
    return V3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
}
inline V3 SphericalDirection(Float sinTheta, Float cosTheta, Float phi, const V3 &x, const V3 &y, const V3 &z) {
    return sinTheta * cosf(phi) * x + sinTheta * sinf(phi) * y + cosTheta * z;
}
inline bool SameHemisphere(const V3 &w, const V3 &wp) { return w.z * wp.z > 0; }

/* geometry.h:1566-1581 */
inline V2 ConcentricSampleDisk(const V2 &u) {
    V2 uOffset = 2.f * u - V2(1, 1);
    if (uOffset.x == 0 && uOffset.y == 0) return V2(0, 0);
    Float theta, r;
    if (Absf(uOffset.x) > Absf(uOffset.y)) {
        r = uOffset.x;
        theta = ORC_PiOver4 * (uOffset.y / uOffset.x); /* double product, narrowed */
    } else {
        r = uOffset.y;
        theta = ORC_PiOver2 - ORC_PiOver4 * (uOffset.x / uOffset.y);
    }
    return r * V2(cosf(theta), sinf(theta));
}
/* geometry.h:1583-1588 */
inline V3 CosineSampleHemisphere(const V2 &u) {
    V2 d = ConcentricSampleDisk(u);
    Float z = sqrtf(Max((Float)0, 1 - d.x * d.x - d.y * d.y));
    return V3(d.x, d.y, z);
}
/* geometry.h:1593-1597 */
inline Float PowerHeuristic(int nf, Float fPdf, int ng, Float gPdf) {
    Float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}
/* geometry.h:1599-1604 */
inline V3 SampleSphere(const V2 &u) {
    Float usqrt = 2 * sqrtf(u[1] * (1 - u[1]));
    Float utheta = 2 * ORC_Pi * u[0];
    return V3(cosf(utheta) * usqrt, sinf(utheta) * usqrt, 1 - 2 * u[1]);
}
inline Float UniformConePdf(Float cosThetaMax) { return 1 / (2 * ORC_Pi * (1 - cosThetaMax)); }
/* geometry.h:1611-1615 */
inline V2 UniformSampleTriangle(const V2 &u) {
    Float su0 = sqrtf(u[0]);
    return V2(1 - su0, u[1] * su0);
}

/* geometry.h:1617-1645, shading-frame trigonometry */
inline Float CosTheta(const V3 &w) { return w.z; }
inline Float Cos2Theta(const V3 &w) { return w.z * w.z; }
inline Float AbsCosTheta(const V3 &w) { return Absf(w.z); }
inline Float Sin2Theta(const V3 &w) { return Max((Float)0, (Float)1 - Cos2Theta(w)); }
inline Float SinTheta(const V3 &w) { return sqrtf(Sin2Theta(w)); }
inline Float TanTheta(const V3 &w) { return SinTheta(w) / CosTheta(w); }
inline Float Tan2Theta(const V3 &w) { return Sin2Theta(w) / Cos2Theta(w); }
inline Float CosPhi(const V3 &w) { Float s = SinTheta(w); return (s == 0) ? 1 : Clampf(w.x / s, -1, 1); }
inline Float SinPhi(const V3 &w) { Float s = SinTheta(w); return (s == 0) ? 0 : Clampf(w.y / s, -1, 1); }
inline Float Cos2Phi(const V3 &w) { return CosPhi(w) * CosPhi(w); }
inline Float Sin2Phi(const V3 &w) { return SinPhi(w) * SinPhi(w); }
/* geometry.h:1647-1660 */
inline V3 Reflect(const V3 &wo, const V3 &n) { return -wo + 2 * Dot(wo, n) * n; }
inline bool Refract(const V3 &wi, const V3 &n, Float eta, V3 *wt) {
    Float cosThetaI = Dot(n, wi);
    Float sin2ThetaI = Max(Float(0), Float(1 - cosThetaI * cosThetaI));
    Float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    Float cosThetaT = sqrtf(1 - sin2ThetaT);
    *wt = eta * -wi + (eta * cosThetaI - cosThetaT) * n;
    return true;
}

/* ---- transform.h ------------------------------------------------------------------- */
struct Xf {
    Float m[4][4], mInv[4][4];
};
inline Xf XfFromFlat(const float *m16, const float *mi16) {
    Xf t;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) { t.m[r][c] = m16[r * 4 + c]; t.mInv[r][c] = mi16[r * 4 + c]; }
    return t;
}
inline Xf XfIdentity() { /* Transform(), transform.h */
    Xf t;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) t.m[r][c] = t.mInv[r][c] = r == c ? 1.f : 0.f;
    return t;
}
inline Xf XfInverse(const Xf &t) { /* transform.h:63-65 */
    Xf r;
    memcpy(r.m, t.mInv, sizeof(r.m));
    memcpy(r.mInv, t.m, sizeof(r.m));
    return r;
}
inline Xf XfTranslate(const V3 &d) { /* transform.cpp:92-98 */
    Xf t;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) t.m[r][c] = t.mInv[r][c] = (r == c) ? 1.f : 0.f;
    t.m[0][3] = d.x; t.m[1][3] = d.y; t.m[2][3] = d.z;
    t.mInv[0][3] = -d.x; t.mInv[1][3] = -d.y; t.mInv[2][3] = -d.z;
    return t;
}
/* transform.h:135-148 */
inline V3 xf_point(const Xf &t, const V3 &p) {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = t.m[0][0] * x + t.m[0][1] * y + t.m[0][2] * z + t.m[0][3];
    Float yp = t.m[1][0] * x + t.m[1][1] * y + t.m[1][2] * z + t.m[1][3];
    Float zp = t.m[2][0] * x + t.m[2][1] * y + t.m[2][2] * z + t.m[2][3];
    Float wp = t.m[3][0] * x + t.m[3][1] * y + t.m[3][2] * z + t.m[3][3];
    if (wp == 1) return V3(xp, yp, zp);
    return V3(xp, yp, zp) / wp;
}
/* transform.h:150-156 */
inline V3 xf_vec(const Xf &t, const V3 &v) {
    Float x = v.x, y = v.y, z = v.z;
    return V3(t.m[0][0] * x + t.m[0][1] * y + t.m[0][2] * z, t.m[1][0] * x + t.m[1][1] * y + t.m[1][2] * z,
              t.m[2][0] * x + t.m[2][1] * y + t.m[2][2] * z);
}
/* transform.h:158-164: normals use the transposed inverse */
inline V3 xf_normal(const Xf &t, const V3 &n) {
    Float x = n.x, y = n.y, z = n.z;
    return V3(t.mInv[0][0] * x + t.mInv[1][0] * y + t.mInv[2][0] * z,
              t.mInv[0][1] * x + t.mInv[1][1] * y + t.mInv[2][1] * z,
              t.mInv[0][2] * x + t.mInv[1][2] * y + t.mInv[2][2] * z);
}
/* transform.h:192-213: point with absolute error bound */
inline V3 xf_point_err(const Xf &t, const V3 &p, V3 *pError) {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = (t.m[0][0] * x + t.m[0][1] * y) + (t.m[0][2] * z + t.m[0][3]);
    Float yp = (t.m[1][0] * x + t.m[1][1] * y) + (t.m[1][2] * z + t.m[1][3]);
    Float zp = (t.m[2][0] * x + t.m[2][1] * y) + (t.m[2][2] * z + t.m[2][3]);
    Float wp = (t.m[3][0] * x + t.m[3][1] * y) + (t.m[3][2] * z + t.m[3][3]);
    Float xAbsSum = (Absf(t.m[0][0] * x) + Absf(t.m[0][1] * y) + Absf(t.m[0][2] * z) + Absf(t.m[0][3]));
    Float yAbsSum = (Absf(t.m[1][0] * x) + Absf(t.m[1][1] * y) + Absf(t.m[1][2] * z) + Absf(t.m[1][3]));
    Float zAbsSum = (Absf(t.m[2][0] * x) + Absf(t.m[2][1] * y) + Absf(t.m[2][2] * z) + Absf(t.m[2][3]));
    *pError = gammaf(3) * V3(xAbsSum, yAbsSum, zAbsSum);
    if (wp == 1) return V3(xp, yp, zp);
    return V3(xp, yp, zp) / wp;
}
/* transform.h:215-248: point with incoming error */
inline V3 xf_point_err2(const Xf &t, const V3 &pt, const V3 &ptError, V3 *absError) {
    Float x = pt.x, y = pt.y, z = pt.z;
    Float xp = (t.m[0][0] * x + t.m[0][1] * y) + (t.m[0][2] * z + t.m[0][3]);
    Float yp = (t.m[1][0] * x + t.m[1][1] * y) + (t.m[1][2] * z + t.m[1][3]);
    Float zp = (t.m[2][0] * x + t.m[2][1] * y) + (t.m[2][2] * z + t.m[2][3]);
    Float wp = (t.m[3][0] * x + t.m[3][1] * y) + (t.m[3][2] * z + t.m[3][3]);
    for (int r = 0; r < 3; r++) {
        (*absError)[r] = (gammaf(3) + (Float)1) * (Absf(t.m[r][0]) * ptError.x + Absf(t.m[r][1]) * ptError.y +
                                                   Absf(t.m[r][2]) * ptError.z) +
                         gammaf(3) * (Absf(t.m[r][0] * x) + Absf(t.m[r][1] * y) + Absf(t.m[r][2] * z) +
                                      Absf(t.m[r][3]));
    }
    if (wp == 1.) return V3(xp, yp, zp);
    return V3(xp, yp, zp) / wp;
}
/* transform.h:250-265 */
inline V3 xf_vec_err(const Xf &t, const V3 &v, V3 *absError) {
    Float x = v.x, y = v.y, z = v.z;
    for (int r = 0; r < 3; r++)
        (*absError)[r] = gammaf(3) * (Absf(t.m[r][0] * v.x) + Absf(t.m[r][1] * v.y) + Absf(t.m[r][2] * v.z));
    return V3(t.m[0][0] * x + t.m[0][1] * y + t.m[0][2] * z, t.m[1][0] * x + t.m[1][1] * y + t.m[1][2] * z,
              t.m[2][0] * x + t.m[2][1] * y + t.m[2][2] * z);
}
/* transform.h:295-307: ray with error bounds; tMax is NOT shortened */
inline Ray xf_ray_err(const Xf &t, const Ray &r, V3 *oError, V3 *dError) {
    V3 o = xf_point_err(t, r.o, oError);
    V3 d = xf_vec_err(t, r.d, dError);
    Float tMax = r.tMax;
    Float lengthSquared = LengthSquared(d);
    if (lengthSquared > 0) {
        Float dt = Dot(Abs(d), *oError) / lengthSquared;
        o += d * dt;
    }
    return Ray(o, d, tMax, r.time, r.medium);
}
/* transform.cpp:258-269 */
inline Bounds3 xf_bounds(const Xf &M, const Bounds3 &b) {
    Bounds3 ret = Bounds3::point(xf_point(M, V3(b.pMin.x, b.pMin.y, b.pMin.z)));
    ret = Union(ret, xf_point(M, V3(b.pMax.x, b.pMin.y, b.pMin.z)));
    ret = Union(ret, xf_point(M, V3(b.pMin.x, b.pMax.y, b.pMin.z)));
    ret = Union(ret, xf_point(M, V3(b.pMin.x, b.pMin.y, b.pMax.z)));
    ret = Union(ret, xf_point(M, V3(b.pMin.x, b.pMax.y, b.pMax.z)));
    ret = Union(ret, xf_point(M, V3(b.pMax.x, b.pMax.y, b.pMin.z)));
    ret = Union(ret, xf_point(M, V3(b.pMax.x, b.pMin.y, b.pMax.z)));
    ret = Union(ret, xf_point(M, V3(b.pMax.x, b.pMax.y, b.pMax.z)));
    return ret;
}
/* transform.cpp:203-208 */
inline bool XfSwapsHandedness(const Xf &t) {
    Float det = t.m[0][0] * (t.m[1][1] * t.m[2][2] - t.m[1][2] * t.m[2][1]) -
                t.m[0][1] * (t.m[1][0] * t.m[2][2] - t.m[1][2] * t.m[2][0]) +
                t.m[0][2] * (t.m[1][0] * t.m[2][1] - t.m[1][1] * t.m[2][0]);
    return det < 0;
}

/* sampling.h:4-17 */
template <typename Pred> inline int FindInterval(int size, const Pred &pred) {
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (pred(middle)) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    return Clampi(first - 1, 0, size - 2);
}

} // namespace orc
