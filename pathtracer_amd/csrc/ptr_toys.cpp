/*
 * ptr_toys.cpp — host-side preparation of the origami signed-distance toys
 * (ProceduralComponent ids 0-3: SDF::OrigamiBoat / Dragon / Whale / Bird, procedural.cpp:402-574).
 *
 * A toy is a handful of triangles whose corners are fixed fractions of the component's half
 * extents.  The reference rebuilds the corners (double products narrowed to float) and, inside
 * SDF::Triangle (procedural.cpp:112-130), the edge vectors, the normal and three double-precision
 * cross products on EVERY map evaluation - up to 256 steps x 21 triangles per ray plus six more
 * evaluations per normal.  None of that depends on the query point, so it is done once per
 * primitive here, with the reference's arithmetic, and the device reads finished DToyTri records
 * (dev_scene.h): per evaluation it is left with three subtractions, three dot products and the
 * distance formula.
 */
#include "ptr_host.h"
#include <cmath>

namespace {

/* component = (float)((double)half * scale + offset) */
struct Term { double scale, offset; };
struct Corner { Term x, y, z; };
struct Face { unsigned char a, b, c, id; };

#define S3(fx, fy, fz) {{fx, 0.0}, {fy, 0.0}, {fz, 0.0}}

/* id 0, procedural.cpp:436-460; 0.03f and 0.01f are float literals in the reference */
const Corner kBoat[6] = {
    /* Poo */ {{0.0, 0.0}, {-0.9, 0.0}, {0.0, 0.0}},
    /* P0  */ S3(0.1, -0.9, 0.6),
    /* P1  */ {{0.1, 0.0}, {-0.9, 0.0}, {0.01, (double)0.03f}},
    /* P2  */ {{0.0, (double)0.01f}, {0.95, 0.0}, {0.0, (double)(0.01f + 0.03f)}},
    /* Pk  */ {{0.95, 0.0}, {0.0, 0.0}, {0.01, (double)0.03f}},
    /* Pn  */ {{0.0, (double)0.01f}, {0.1, 0.0}, {0.9, 0.0}},
};
const Face kBoatFaces[3] = {{1, 2, 3, 0}, {1, 4, 0, 1}, {1, 4, 5, 2}};

/* id 3, procedural.cpp:402-434 */
const Corner kBird[10] = {S3(-0.25, 0.88, 0.30), S3(0.05, 0.50, 0.13),   S3(-0.15, 0.05, 0.13),  S3(0.60, -0.10, 0.08),  S3(0.50, 0.02, 0.03),
                          S3(0.75, 0.20, 0.00),  S3(0.85, 0.00, 0.00),   S3(-0.20, -0.50, 0.03), S3(-0.85, -0.90, 0.00), S3(-0.25, -0.80, 0.00)};
const Face kBirdFaces[6] = {{0, 1, 2, 0}, {1, 2, 3, 1}, {3, 4, 5, 2}, {6, 5, 3, 1}, {2, 3, 7, 1}, {2, 8, 9, 2}};

/* id 2, procedural.cpp:462-503 */
const Corner kWhale[13] = {S3(-0.15, 0.99, 0.20), S3(-0.65, -0.05, 0.80), S3(-0.10, -0.80, 0.99), S3(-0.98, 0.82, 0.40),  S3(0.98, -0.15, 0.03),
                           S3(0.90, -0.80, 0.70), S3(-0.98, 0.15, 0.50),  S3(0.85, -0.99, 0.80),  S3(-0.80, -0.99, 0.80), S3(-0.96, -0.76, 0.70),
                           S3(0.95, 0.85, 0.00),  S3(0.65, 0.10, 0.00),   S3(0.75, 0.05, 0.03)};
const Face kWhaleFaces[9] = {{0, 1, 2, 0}, {0, 3, 4, 1}, {3, 4, 5, 1}, {3, 5, 6, 1}, {7, 8, 9, 2}, {7, 6, 9, 2}, {7, 6, 5, 2}, {4, 12, 10, 1}, {12, 10, 11, 0}};

/* id 1, procedural.cpp:505-574 (every triangle reports shape id 5, :572) */
enum { P0, P1, P2, P3, P4, P5Z, P6Z, P7Z, P8Z, P2S, P9, P10, P11, P12, P13, P14, P15, Q3, Q2, Q0, Q6, Q4, Q5, Q7, Q8, Q9, T0, T1, T2, T3, T4, T5 };
const Corner kDragon[32] = {
    /* P0  */ S3(-0.70, 0.85, 0.05),  /* P1  */ S3(-0.25, 0.40, 0.06),  /* P2  */ S3(-0.55, -0.15, 0.05), /* P3  */ S3(0.10, -0.60, 0.09),
    /* P4  */ S3(0.20, 0.15, 0.08),   /* P5Z */ S3(0.05, 0.25, 0.00),   /* P6Z */ S3(0.35, 0.10, 0.02),   /* P7Z */ S3(0.45, -0.60, 0.05),
    /* P8Z */ S3(-0.35, -0.60, 0.05), /* P2S */ S3(-0.55, -0.15, 0.05), /* P9  */ S3(0.40, 0.20, 0.00),   /* P10 */ S3(0.70, -0.75, 0.10),
    /* P11 */ S3(0.43, -0.96, 0.15),  /* P12 */ S3(-0.65, -0.96, 0.15), /* P13 */ S3(-0.45, -0.05, 0.05), /* P14 */ S3(0.80, 0.10, 0.01),
    /* P15 */ S3(0.50, -0.60, 0.01),  /* Q3  */ S3(0.60, 0.10, 0.01),   /* Q2  */ S3(0.55, 0.85, 0.03),   /* Q0  */ S3(0.70, 0.90, 0.03),
    /* Q6  */ S3(0.71, 0.70, 0.10),   /* Q4  */ S3(0.90, 0.90, 0.00),   /* Q5  */ S3(0.88, 0.80, 0.00),   /* Q7  */ S3(0.65, 1.00, 0.00),
    /* Q8  */ S3(0.45, 1.00, 0.00),   /* Q9  */ S3(0.64, 0.88, 0.03),   /* T0  */ S3(-0.85, -0.25, 0.03), /* T1  */ S3(-0.80, -0.50, 0.03),
    /* T2  */ S3(-0.83, 0.16, 0.01),  /* T3  */ S3(-1.00, 0.19, 0.01),  /* T4  */ S3(-0.53, -0.23, 0.01), /* T5  */ S3(-0.90, 0.75, 0.00)};
const Face kDragonFaces[21] = {{P0, P1, P2, 5},   {P1, P2, P3, 5},   {P1, P3, P4, 5},  {P5Z, P6Z, P7Z, 5}, {P5Z, P7Z, P8Z, 5}, {P5Z, P8Z, P2S, 5}, {P9, P6Z, P10, 5},
                               {P6Z, P11, P10, 5}, {P8Z, P12, P13, 5}, {P15, P14, P6Z, 5}, {Q3, Q2, P14, 5},   {Q0, P14, Q2, 5},   {Q6, Q5, Q0, 5},    {Q5, Q4, Q0, 5},
                               {Q4, Q7, Q9, 5},   {Q7, Q9, Q8, 5},   {P9, T0, T1, 5},  {T1, P9, P7Z, 5},   {T0, T2, T3, 5},    {T0, T2, T4, 5},    {T3, T2, T5, 5}};
#undef S3

struct V { float x, y, z; };
V sub(const V &a, const V &b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; }
V cross_d(const V &a, const V &b) { /* geometry.h:642-648: in double, narrowed per component */
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V{(float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx))};
}
float dot2(const V &v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
void put(float *d, const V &v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

V corner_at(const float *half, const Corner &k) {
    return V{(float)((double)half[0] * k.x.scale + k.x.offset), (float)((double)half[1] * k.y.scale + k.y.offset),
             (float)((double)half[2] * k.z.scale + k.z.offset)};
}

} // namespace

/* Appends the triangle records of toy `id` for a component with the given half extents; returns
 * how many were added (0: not an origami toy). */
int ptr_build_toy(int id, const float *half, std::vector<DToyTri> *out) {
    const Corner *corners;
    const Face *faces;
    int n;
    switch (id) {
    case 0: corners = kBoat; faces = kBoatFaces; n = 3; break;
    case 1: corners = kDragon; faces = kDragonFaces; n = 21; break;
    case 2: corners = kWhale; faces = kWhaleFaces; n = 9; break;
    case 3: corners = kBird; faces = kBirdFaces; n = 6; break;
    default: return 0;
    }
    for (int i = 0; i < n; i++) {
        const V a = corner_at(half, corners[faces[i].a]), b = corner_at(half, corners[faces[i].b]), c = corner_at(half, corners[faces[i].c]);
        const V ba = sub(b, a), cb = sub(c, b), ac = sub(a, c);
        const V nor = cross_d(ba, ac);
        DToyTri t;
        put(t.a, a); put(t.b, b); put(t.c, c);
        put(t.ba, ba); put(t.cb, cb); put(t.ac, ac);
        put(t.nor, nor);
        put(t.e0, cross_d(ba, nor)); put(t.e1, cross_d(cb, nor)); put(t.e2, cross_d(ac, nor));
        t.d2ba = dot2(ba); t.d2cb = dot2(cb); t.d2ac = dot2(ac); t.d2nor = dot2(nor);
        t.id = faces[i].id;
        t.pad = 0;
        out->push_back(t);
    }
    return n;
}
