/*
 * pt_math.h — host-side value types of the scene-description API.
 *
 * Mirrors the subset of src/core/geometry.h and src/core/transform.h that scene functions
 * (src/main.cpp:459-1095) touch: vec3f / Point3f / Spectrum / Bounds3f / Transform and the
 * transform factories.  Same names and argument meaning so those scene functions compile
 * unchanged; arithmetic follows the reference (Float matrices, Gauss-Jordan inverse, double
 * Pi in Radians) so a Transform built here equals the one the reference would build.
 * Setup-time only: nothing in here runs per ray.
 */
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>

typedef float Float; /* geometry.h:50 */
#define Infinity FLT_MAX
#define Pi 3.14159265358979323846

inline Float Radians(Float deg) { return (Pi / 180) * deg; } /* geometry.h:88 */
inline Float Absf(Float v) { return v > 0 ? v : -v; }
inline Float Max(Float a, Float b) { return a < b ? b : a; }
inline Float Min(Float a, Float b) { return a < b ? a : b; }

struct vec3f {
    Float x, y, z;
    vec3f() : x(0), y(0), z(0) {}
    vec3f(Float a) : x(a), y(a), z(a) {}
    vec3f(Float a, Float b, Float c) : x(a), y(b), z(c) {}
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    vec3f operator+(const vec3f &v) const { return vec3f(x + v.x, y + v.y, z + v.z); }
    vec3f operator-(const vec3f &v) const { return vec3f(x - v.x, y - v.y, z - v.z); }
    vec3f operator-() const { return vec3f(-x, -y, -z); }
    vec3f operator*(Float s) const { return vec3f(x * s, y * s, z * s); }
    vec3f operator*(const vec3f &v) const { return vec3f(x * v.x, y * v.y, z * v.z); }
    vec3f operator/(Float f) const { Float inv = (Float)1 / f; return vec3f(x * inv, y * inv, z * inv); }
    Float LengthSquared() const { return x * x + y * y + z * z; }
    Float Length() const { return std::sqrt(LengthSquared()); }
};
inline vec3f operator*(Float s, const vec3f &v) { return v * s; }
typedef vec3f Spectrum; /* geometry.h:771 */

struct Point3f {
    Float x, y, z;
    Point3f() : x(0), y(0), z(0) {}
    Point3f(Float a) : x(a), y(a), z(a) {}
    Point3f(Float a, Float b, Float c) : x(a), y(b), z(c) {}
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    vec3f operator-(const Point3f &p) const { return vec3f(x - p.x, y - p.y, z - p.z); }
    Point3f operator+(const vec3f &v) const { return Point3f(x + v.x, y + v.y, z + v.z); }
    Point3f operator-(const vec3f &v) const { return Point3f(x - v.x, y - v.y, z - v.z); }
    Point3f operator/(Float f) const { Float inv = (Float)1 / f; return Point3f(x * inv, y * inv, z * inv); }
};
struct Point2f {
    Float x, y;
    Point2f() : x(0), y(0) {}
    Point2f(Float a, Float b) : x(a), y(b) {}
};
struct vec2f {
    Float x, y;
    vec2f() : x(0), y(0) {}
    vec2f(Float a, Float b) : x(a), y(b) {}
};
struct Point3i {
    int x, y, z;
    Point3i() : x(0), y(0), z(0) {}
    Point3i(int a, int b, int c) : x(a), y(b), z(c) {}
};
typedef vec3f Normal3f;

inline vec3f ToVec3(const Point3f &p) { return vec3f(p.x, p.y, p.z); }
inline Float Dot(const vec3f &a, const vec3f &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3f Normalize(const vec3f &v) { return v / v.Length(); }
/* geometry.h:642-648: double products */
inline vec3f Cross(const vec3f &a, const vec3f &b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return vec3f((Float)((ay * bz) - (az * by)), (Float)((az * bx) - (ax * bz)), (Float)((ax * by) - (ay * bx)));
}

struct Bounds3f { /* geometry.h:1310-1327 */
    Point3f pMin, pMax;
    Bounds3f() : pMin(FLT_MAX, FLT_MAX, FLT_MAX), pMax(FLT_MIN, FLT_MIN, FLT_MIN) {}
    Bounds3f(const Point3f &a, const Point3f &b)
        : pMin(Min(a.x, b.x), Min(a.y, b.y), Min(a.z, b.z)), pMax(Max(a.x, b.x), Max(a.y, b.y), Max(a.z, b.z)) {}
};

struct Matrix4x4 { /* transform.h:8-45 */
    Float m[4][4];
    Matrix4x4() {
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) m[r][c] = (r == c) ? 1.f : 0.f;
    }
    Matrix4x4(Float t00, Float t01, Float t02, Float t03, Float t10, Float t11, Float t12, Float t13, Float t20, Float t21,
              Float t22, Float t23, Float t30, Float t31, Float t32, Float t33) {
        Float v[16] = {t00, t01, t02, t03, t10, t11, t12, t13, t20, t21, t22, t23, t30, t31, t32, t33};
        memcpy(m, v, sizeof(v));
    }
    static Matrix4x4 Mul(const Matrix4x4 &a, const Matrix4x4 &b) {
        Matrix4x4 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
        return r;
    }
};
Matrix4x4 Transpose(const Matrix4x4 &m);
Matrix4x4 Inverse(const Matrix4x4 &m);

class Transform { /* transform.h:47-117 */
  public:
    Matrix4x4 m, mInv;
    Transform() {}
    Transform(const Matrix4x4 &mm) : m(mm), mInv(Inverse(mm)) {}
    Transform(const Matrix4x4 &mm, const Matrix4x4 &mi) : m(mm), mInv(mi) {}
    friend Transform Inverse(const Transform &t) { return Transform(t.mInv, t.m); }
    Transform operator*(const Transform &t2) const { return Transform(Matrix4x4::Mul(m, t2.m), Matrix4x4::Mul(t2.mInv, mInv)); }
    Point3f operator()(const Point3f &p) const; /* transform.h:135-148 */
    vec3f operator()(const vec3f &v) const;
};

Transform Translate(const vec3f &delta);
Transform Translate(Float x, Float y, Float z);
Transform Translate(Float u);
Transform Scale(Float x, Float y, Float z);
Transform Scale(Float u);
Transform RotateX(Float theta);
Transform RotateY(Float theta);
Transform RotateZ(Float theta);
Transform Rotate(Float theta, const vec3f &axis);
