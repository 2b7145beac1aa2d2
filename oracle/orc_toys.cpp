/*
 * orc_toys.cpp — ORACLE (test infrastructure only).  PARITY UNPINNED.
 * ProceduralComponent::Map dispatch (src/core/procedural.cpp:380-400) and the signed-distance
 * maps behind it: SDF::Terrain (:398-400), SDF::Triangle (:112-130) with its helpers Sign / Dot2
 * (:6-18), and the four origami toys (:402-574).  A toy is a set of triangles whose corners are
 * fixed fractions of the component's half extents, mirrored in z (the boat also in x); the
 * reference writes each corner as `±half.k * <double literal>` (a double product narrowed into
 * the float vector) and evaluates every triangle's distance on every map call.  Restated here as
 * corner tables + one evaluation loop, same arithmetic per corner and per triangle.
 */
#include "orc_scene.h"
#include <cstdio>
#include <cstdlib>

namespace orc {

static Float ExtentOn(const Bounds3 &b, int i) { /* geometry.h:1357-1362 */
    if (i == 0) return Absf(b.pMax.x - b.pMin.x);
    if (i == 1) return Absf(b.pMax.y - b.pMin.y);
    return Absf(b.pMax.z - b.pMin.z);
}

static Float SignF(Float x) { /* procedural.cpp:14-18 */
    if (IsZero(x)) return 0;
    if (x > 0) return 1;
    return -1;
}
static Float Dot2(const V3 &v) { return Dot(v, v); } /* :6-8 */
static Float Clamp01d(Float v) { /* Clamp(float, 0.0, 1.0): geometry.h:173-178, result narrowed to the float argument type */
    if (v < 0.0) return (Float)0.0;
    if (v > 1.0) return (Float)1.0;
    return v;
}

/* SDF::Triangle, procedural.cpp:112-130 */
static Float SDFTriangle(const V3 &p, const V3 &a, const V3 &b, const V3 &c) {
    V3 ba = b - a, pa = p - a;
    V3 cb = c - b, pb = p - b;
    V3 ac = a - c, pc = p - c;
    V3 nor = Cross(ba, ac);
    Float sa = SignF(Dot(Cross(ba, nor), pa));
    Float sb = SignF(Dot(Cross(cb, nor), pb));
    Float sc = SignF(Dot(Cross(ac, nor), pc));
    Float ss = sa + sb + sc;
    if (ss < 2) {
        Float a2 = Dot2(ba * Clamp01d(Dot(ba, pa) / Dot2(ba)) - pa);
        Float b2 = Dot2(cb * Clamp01d(Dot(cb, pb) / Dot2(cb)) - pb);
        Float c2 = Dot2(ac * Clamp01d(Dot(ac, pc) / Dot2(ac)) - pc);
        return sqrtf(Min(a2, Min(b2, c2)));
    }
    return sqrtf(Dot(nor, pa) * Dot(nor, pa) / Dot2(nor));
}

/* a corner `(sx*half.x*fx, sy*half.y*fy, half.z*fz)`: each component is a double product narrowed to Float */
struct Corner { double fx, fy, fz; };
static V3 At(const V3 &half, const Corner &k) { return V3((Float)(half.x * k.fx), (Float)(half.y * k.fy), (Float)(half.z * k.fz)); }
struct Face { int a, b, c, id; };

static Float FoldedMin(const V3 &q, const V3 &half, const Corner *corners, const Face *faces, int nFaces, int *firstMinId) {
    Float best = ORC_Infinity;
    int id = -1;
    for (int i = 0; i < nFaces; i++) {
        Float d = SDFTriangle(q, At(half, corners[faces[i].a]), At(half, corners[faces[i].b]), At(half, corners[faces[i].c]));
        if (best > d) { best = d; id = faces[i].id; }
    }
    *firstMinId = id;
    return best;
}

/* :402-434 */
static const Corner kBird[10] = {{-0.25, 0.88, 0.30}, {0.05, 0.50, 0.13}, {-0.15, 0.05, 0.13}, {0.60, -0.10, 0.08}, {0.50, 0.02, 0.03},
                                 {0.75, 0.20, 0.00},  {0.85, 0.00, 0.00}, {-0.20, -0.50, 0.03}, {-0.85, -0.90, 0.00}, {-0.25, -0.80, 0.00}};
static const Face kBirdFaces[6] = {{0, 1, 2, 0}, {1, 2, 3, 1}, {3, 4, 5, 2}, {6, 5, 3, 1}, {2, 3, 7, 1}, {2, 8, 9, 2}};

/* :462-503 */
static const Corner kWhale[13] = {{-0.15, 0.99, 0.20}, {-0.65, -0.05, 0.80}, {-0.10, -0.80, 0.99}, {-0.98, 0.82, 0.40}, {0.98, -0.15, 0.03},
                                  {0.90, -0.80, 0.70}, {-0.98, 0.15, 0.50},  {0.85, -0.99, 0.80},  {-0.80, -0.99, 0.80}, {-0.96, -0.76, 0.70},
                                  {0.95, 0.85, 0.00},  {0.65, 0.10, 0.00},   {0.75, 0.05, 0.03}};
static const Face kWhaleFaces[9] = {{0, 1, 2, 0}, {0, 3, 4, 1}, {3, 4, 5, 1}, {3, 5, 6, 1}, {7, 8, 9, 2},
                                    {7, 6, 9, 2}, {7, 6, 5, 2}, {4, 12, 10, 1}, {12, 10, 11, 0}};

/* :505-574; corner order: P0 P1 P2 P3 P4 P5Zp P6Zp P7Zp P8Zp P2Sh P9pp P10pp P11pp P12m P13m P14p P15p Q3 Q2 Q0 Q6 Q4 Q5 Q7 Q8 Q9 T0..T5 */
static const Corner kDragon[32] = {
    {-0.70, 0.85, 0.05},  {-0.25, 0.40, 0.06},  {-0.55, -0.15, 0.05}, {0.10, -0.60, 0.09},  {0.20, 0.15, 0.08},  {0.05, 0.25, 0.00},
    {0.35, 0.10, 0.02},   {0.45, -0.60, 0.05},  {-0.35, -0.60, 0.05}, {-0.55, -0.15, 0.05}, {0.40, 0.20, 0.00},  {0.70, -0.75, 0.10},
    {0.43, -0.96, 0.15},  {-0.65, -0.96, 0.15}, {-0.45, -0.05, 0.05}, {0.80, 0.10, 0.01},   {0.50, -0.60, 0.01}, {0.60, 0.10, 0.01},
    {0.55, 0.85, 0.03},   {0.70, 0.90, 0.03},   {0.71, 0.70, 0.10},   {0.90, 0.90, 0.00},   {0.88, 0.80, 0.00},  {0.65, 1.00, 0.00},
    {0.45, 1.00, 0.00},   {0.64, 0.88, 0.03},   {-0.85, -0.25, 0.03}, {-0.80, -0.50, 0.03}, {-0.83, 0.16, 0.01}, {-1.00, 0.19, 0.01},
    {-0.53, -0.23, 0.01}, {-0.90, 0.75, 0.00}};
enum { dP0, dP1, dP2, dP3, dP4, dP5, dP6, dP7, dP8, dP2S, dP9, dP10, dP11, dP12, dP13, dP14, dP15, dQ3, dQ2, dQ0, dQ6, dQ4, dQ5, dQ7, dQ8, dQ9,
       dT0, dT1, dT2, dT3, dT4, dT5 };
static const Face kDragonFaces[21] = {
    {dP0, dP1, dP2, 5},   {dP1, dP2, dP3, 5},   {dP1, dP3, dP4, 5},  {dP5, dP6, dP7, 5},  {dP5, dP7, dP8, 5},  {dP5, dP8, dP2S, 5}, {dP9, dP6, dP10, 5},
    {dP6, dP11, dP10, 5}, {dP8, dP12, dP13, 5}, {dP15, dP14, dP6, 5}, {dQ3, dQ2, dP14, 5}, {dQ0, dP14, dQ2, 5}, {dQ6, dQ5, dQ0, 5},  {dQ5, dQ4, dQ0, 5},
    {dQ4, dQ7, dQ9, 5},   {dQ7, dQ9, dQ8, 5},   {dP9, dT0, dT1, 5},  {dT1, dP9, dP7, 5},  {dT0, dT2, dT3, 5},  {dT0, dT2, dT4, 5},  {dT3, dT2, dT5, 5}};

/* :436-460: some corners carry absolute offsets (`half.z * 0.01 + 0.03f`, `0.01f`), so they are written out */
static Float Boat(const V3 &p, const V3 &half, int *shapeId) {
    V3 q(Absf(p.x), p.y, Absf(p.z));
    V3 Poo(0.f, (Float)(-half.y * 0.9), 0.f);
    V3 P0((Float)(half.x * 0.1), (Float)(-half.y * 0.9), (Float)(half.z * 0.6));
    V3 P1((Float)(half.x * 0.1), (Float)(-half.y * 0.9), (Float)(half.z * 0.01 + 0.03f));
    V3 P2(0.01f, (Float)(half.y * 0.95), 0.01f + 0.03f);
    V3 Pk((Float)(half.x * 0.95), 0.f, (Float)(half.z * 0.01 + 0.03f));
    V3 Pn(0.01f, (Float)(half.y * 0.1), (Float)(half.z * 0.9));
    Float d[3] = {SDFTriangle(q, P0, P1, P2), SDFTriangle(q, P0, Pk, Poo), SDFTriangle(q, P0, Pk, Pn)};
    Float best = ORC_Infinity;
    int id = -1;
    for (int i = 0; i < 3; i++)
        if (best > d[i]) { best = d[i]; id = i; }
    *shapeId = id;
    return best;
}

Float SDFMapToy(int id, const Bounds3 &bounds, const V3 &p, int *shapeId) {
    V3 length(ExtentOn(bounds, 0), ExtentOn(bounds, 1), ExtentOn(bounds, 2));
    V3 half = length * 0.5;
    V3 q(p.x, p.y, Absf(p.z));
    switch (id) {
    case 0: return Boat(p, half, shapeId);
    case 1: { /* the dragon reports one shape id for all of its triangles (:572) */
        int ignored;
        Float d = FoldedMin(q, half, kDragon, kDragonFaces, 21, &ignored);
        *shapeId = 5;
        return d;
    }
    case 2: return FoldedMin(q, half, kWhale, kWhaleFaces, 9, shapeId);
    case 3: return FoldedMin(q, half, kBird, kBirdFaces, 6, shapeId);
    case 4: { /* SDF::Terrain == SDF::Box(q, half), procedural.cpp:398-400 */
        V3 d = Abs(p) - half;
        return Min(Max(d.x, Max(d.y, d.z)), 0.0) + Length(V3(Max(d.x, 0.f), Max(d.y, 0.f), Max(d.z, 0.f)));
    }
    default: /* "Unknown component id": Map returns false == 0 (:389-394) */
        return 0;
    }
}

} // namespace orc
