/*
 * dev_isect.h — ray/shape intersection and BVH traversal on gfx950.
 *
 * Two halves per shape, split the way a wavefront pipeline wants them:
 *   *_hit_t   : what the trace kernels need — accept/reject and the hit parameter
 *               (reference: the first half of each Shape::Intersect)
 *   make_si   : what the shade kernel needs — the SurfaceInteraction rebuilt from
 *               (ray, primitive, face, t), i.e. the second half of the same functions.
 * Both halves perform the reference's arithmetic (citations inline), so t, barycentrics, hit
 * point, normal and dpdu agree bit-for-bit with a host build up to the first libm call.
 */
#pragma once
#include "dev_math.h"
#include "dev_scene.h"

struct SI { /* the fields of SurfaceInteraction the hot path reads (interaction.h:9-75) */
    f3 p, pError, wo, n, dpdu;
    f2 uv;
    int prim, face;
    int med_in, med_out;
};

/* geometry.h:1481-1499 with 1/d hoisted (the reference recomputes it per test; same values) and
 * without the early exits: the entry bound only grows and the exit bound only shrinks from axis to
 * axis, so "t0 > t1 after some axis" and "t0 > t1 after the last axis" are the same predicate (NaNs
 * from 0 * inf leave both bounds untouched in either form).  Straight-line code: no EXEC-mask
 * juggling per axis, and the tests of several boxes interleave. */
PD bool slab_test6(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, const f3 &o, const f3 &inv, float tmax, float *t_entry,
                   float *t_exit = nullptr) {
    const float k = 1 + 2 * gammar(3);
    float t0 = 0, t1 = tmax;
    {
        float tn = (mnx - o.x) * inv.x, tf = (mxx - o.x) * inv.x;
        const float lo = tn > tf ? tf : tn, hi = (tn > tf ? tn : tf) * k;
        t0 = lo > t0 ? lo : t0;
        t1 = hi < t1 ? hi : t1;
    }
    {
        float tn = (mny - o.y) * inv.y, tf = (mxy - o.y) * inv.y;
        const float lo = tn > tf ? tf : tn, hi = (tn > tf ? tn : tf) * k;
        t0 = lo > t0 ? lo : t0;
        t1 = hi < t1 ? hi : t1;
    }
    {
        float tn = (mnz - o.z) * inv.z, tf = (mxz - o.z) * inv.z;
        const float lo = tn > tf ? tf : tn, hi = (tn > tf ? tn : tf) * k;
        t0 = lo > t0 ? lo : t0;
        t1 = hi < t1 ? hi : t1;
    }
    *t_entry = t0;
    if (t_exit) *t_exit = t1;
    return !(t0 > t1);
}
PD bool slab_test(const float *bmin, const float *bmax, const f3 &o, const f3 &inv, float tmax, float *t_entry, float *t_exit = nullptr) {
    return slab_test6(bmin[0], bmin[1], bmin[2], bmax[0], bmax[1], bmax[2], o, inv, tmax, t_entry, t_exit);
}

/* The geometry-level form of the test: the same products and the same (1 + 2 gamma(3)) widening,
 * with the swaps and running bounds as hardware min / max (v_min3 / v_max3).  It answers like
 * slab_test6 for every ray except one lying exactly in a face plane of the box with a zero direction
 * component (0 * inf = NaN: the reference's compare-and-select form skips that bound, IEEE min / max
 * take the other operand) - a case where "enters the box" has no defined answer; these boxes only
 * gate which leaves of OUR tree are looked at, the top level keeps the reference's exact form. */
PD bool slab_minmax(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, const f3 &o, const f3 &inv, float tmax, float *t_entry) {
    const float k = 1 + 2 * gammar(3);
    const float ax = (mnx - o.x) * inv.x, bx = (mxx - o.x) * inv.x;
    const float ay = (mny - o.y) * inv.y, by = (mxy - o.y) * inv.y;
    const float az = (mnz - o.z) * inv.z, bz = (mxz - o.z) * inv.z;
    const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
    /* min(hx k, hy k, hz k) = min(hx, hy, hz) k to the bit: rounding is monotonic and k > 0 (NaN and inf pass through alike) */
    const float t1 = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * k, tmax);
    *t_entry = t0;
    return t0 <= t1;
}

/* One visit of a 4-wide node (dev_scene.h Node128): tests the occupied child slots and leaves the
 * refs of those the ray enters in r0..r3, nearest first (BVH_NO_CHILD pads the tail).  A sorting
 * network over (entry distance, ref) pairs keeps everything in registers. */
#define BVH4_CSWAP(ta, ra, tb, rb)                       \
    do {                                                 \
        const bool sw_ = tb < ta;                        \
        const float tl_ = sw_ ? tb : ta, th_ = sw_ ? ta : tb; \
        const int rl_ = sw_ ? rb : ra, rh_ = sw_ ? ra : rb;   \
        ta = tl_; tb = th_; ra = rl_; rb = rh_;          \
    } while (0)
PD int bvh4_visit_loaded(const float4 &a0, const float4 &b0, const float4 &a1, const float4 &b1, const float4 &a2, const float4 &b2, const float4 &a3,
                         const float4 &b3, const f3 &ro, const f3 &inv, float tmax, int &r0, int &r1, int &r2, int &r3, unsigned *tested) {
    r0 = __float_as_int(b0.z); r1 = __float_as_int(b1.z); r2 = __float_as_int(b2.z); r3 = __float_as_int(b3.z);
    float t0 = kInfinity, t1 = kInfinity, t2 = kInfinity, t3 = kInfinity;
    if (tested) *tested += (r0 != BVH_NO_CHILD) + (r1 != BVH_NO_CHILD) + (r2 != BVH_NO_CHILD) + (r3 != BVH_NO_CHILD);
    {
        const bool h0 = slab_minmax(a0.x, a0.y, a0.z, a0.w, b0.x, b0.y, ro, inv, tmax, &t0) & (r0 != BVH_NO_CHILD);
        const bool h1 = slab_minmax(a1.x, a1.y, a1.z, a1.w, b1.x, b1.y, ro, inv, tmax, &t1) & (r1 != BVH_NO_CHILD);
        const bool h2 = slab_minmax(a2.x, a2.y, a2.z, a2.w, b2.x, b2.y, ro, inv, tmax, &t2) & (r2 != BVH_NO_CHILD);
        const bool h3 = slab_minmax(a3.x, a3.y, a3.z, a3.w, b3.x, b3.y, ro, inv, tmax, &t3) & (r3 != BVH_NO_CHILD);
        r0 = h0 ? r0 : BVH_NO_CHILD; t0 = h0 ? t0 : kInfinity;
        r1 = h1 ? r1 : BVH_NO_CHILD; t1 = h1 ? t1 : kInfinity;
        r2 = h2 ? r2 : BVH_NO_CHILD; t2 = h2 ? t2 : kInfinity;
        r3 = h3 ? r3 : BVH_NO_CHILD; t3 = h3 ? t3 : kInfinity;
    }
    BVH4_CSWAP(t0, r0, t1, r1);
    BVH4_CSWAP(t2, r2, t3, r3);
    BVH4_CSWAP(t0, r0, t2, r2);
    BVH4_CSWAP(t1, r1, t3, r3);
    BVH4_CSWAP(t1, r1, t2, r2);
    return (r0 != BVH_NO_CHILD) + (r1 != BVH_NO_CHILD) + (r2 != BVH_NO_CHILD) + (r3 != BVH_NO_CHILD);
}
PD int bvh4_visit(const Node128 *nodes, int node, const f3 &ro, const f3 &inv, float tmax, int &r0, int &r1, int &r2, int &r3, unsigned *tested) {
    const float4 *np = reinterpret_cast<const float4 *>(nodes + node);
    /* child i = np[2 i] (min xyz, max x), np[2 i + 1] (max y, max z, ref, -) */
    const float4 a0 = np[0], b0 = np[1], a1 = np[2], b1 = np[3], a2 = np[4], b2 = np[5], a3 = np[6], b3 = np[7];
    return bvh4_visit_loaded(a0, b0, a1, b1, a2, b2, a3, b3, ro, inv, tmax, r0, r1, r2, r3, tested);
}
/* The visit from the quantised copy of the node (dev_scene.h NodeQ): four 16-byte requests instead of eight, the planes decoded
 * with one fused multiply-add each (the expression ptr_build.cpp quantise_nodes rounded them against) */
PD int bvh4_visit_q(const NodeQ *qnodes, int node, const f3 &ro, const f3 &inv, float tmax, int &r0, int &r1, int &r2, int &r3, unsigned *tested) {
    const float4 *np = reinterpret_cast<const float4 *>(qnodes + node);
    const float4 A = np[0];
    const int4 R = reinterpret_cast<const int4 *>(np)[1];
    const uint4 Q = reinterpret_cast<const uint4 *>(np)[2];
    const uint2 Q2 = *reinterpret_cast<const uint2 *>(np + 3);
    const uint32_t e = __float_as_uint(A.w);
    const float sx = __uint_as_float((e & 0xffu) << 23), sy = __uint_as_float(((e >> 8) & 0xffu) << 23), sz = __uint_as_float(((e >> 16) & 0xffu) << 23);
#define NQ_PLANE(word, c, s, o) __builtin_fmaf((float)(((word) >> (8 * (c))) & 0xffu), s, o)
#define NQ_A(c) make_float4(NQ_PLANE(Q.x, c, sx, A.x), NQ_PLANE(Q.y, c, sy, A.y), NQ_PLANE(Q.z, c, sz, A.z), NQ_PLANE(Q.w, c, sx, A.x))
#define NQ_B(c, ref) make_float4(NQ_PLANE(Q2.x, c, sy, A.y), NQ_PLANE(Q2.y, c, sz, A.z), __int_as_float(ref), 0.f)
    const float4 a0 = NQ_A(0), b0 = NQ_B(0, R.x), a1 = NQ_A(1), b1 = NQ_B(1, R.y), a2 = NQ_A(2), b2 = NQ_B(2, R.z), a3 = NQ_A(3), b3 = NQ_B(3, R.w);
#undef NQ_PLANE
#undef NQ_A
#undef NQ_B
    return bvh4_visit_loaded(a0, b0, a1, b1, a2, b2, a3, b3, ro, inv, tmax, r0, r1, r2, r3, tested);
}
/* The same with nodes[0 .. PT_LDS_TOP) - the upper levels of every geometry tree (BVH_NODE_PREFIX, ptr_build.cpp) - read
 * from the block's LDS copy instead: rows of 9 float4, so that lanes on different nodes spread over the banks */
#ifndef PT_LDS_TOP
#define PT_LDS_TOP 0
#endif
#define PT_LDS_ROW 9
PD int bvh4_visit_staged(const Node128 *nodes, const float4 *staged, int node, const f3 &ro, const f3 &inv, float tmax, int &r0, int &r1, int &r2, int &r3,
                         unsigned *tested) {
#if PT_LDS_TOP > 0
    float4 a0, b0, a1, b1, a2, b2, a3, b3;
    if (node < PT_LDS_TOP) {
        const float4 *lp = staged + node * PT_LDS_ROW;
        a0 = lp[0]; b0 = lp[1]; a1 = lp[2]; b1 = lp[3]; a2 = lp[4]; b2 = lp[5]; a3 = lp[6]; b3 = lp[7];
    } else {
        const float4 *np = reinterpret_cast<const float4 *>(nodes + node);
        a0 = np[0]; b0 = np[1]; a1 = np[2]; b1 = np[3]; a2 = np[4]; b2 = np[5]; a3 = np[6]; b3 = np[7];
    }
    return bvh4_visit_loaded(a0, b0, a1, b1, a2, b2, a3, b3, ro, inv, tmax, r0, r1, r2, r3, tested);
#else
    return bvh4_visit(nodes, node, ro, inv, tmax, r0, r1, r2, r3, tested);
#endif
}
/* block-wide copy of the staged nodes; every thread of the block calls it once, before the first visit */
PD void bvh4_stage(const DScene &sc, float4 *staged) {
#if PT_LDS_TOP > 0
    const int ns = sc.n_nodes < PT_LDS_TOP ? sc.n_nodes : PT_LDS_TOP;
    const float4 *src = reinterpret_cast<const float4 *>(sc.nodes);
    for (int i = threadIdx.x; i < ns * 8; i += blockDim.x) staged[(i >> 3) * PT_LDS_ROW + (i & 7)] = src[i];
    __syncthreads();
#endif
}

/* ---- triangle, mesh.cpp:162-249 (watertight test, double fallback for zero edge functions) ----
 * The permutation and shear depend on the ray only (mesh.cpp:176-186); the reference recomputes
 * them for every triangle, here they are set up once per ray - same operations, same values. */
struct TriRay {
    int kx, ky, kz;
    float Sx, Sy, Sz;
};
PD TriRay tri_ray_setup(const f3 &rd) {
    TriRay tr;
    tr.kz = max_dim(abs3(rd));
    tr.kx = tr.kz + 1; if (tr.kx == 3) tr.kx = 0;
    tr.ky = tr.kx + 1; if (tr.ky == 3) tr.ky = 0;
    f3 d(comp(rd, tr.kx), comp(rd, tr.ky), comp(rd, tr.kz));
    tr.Sx = -d.x / d.z; tr.Sy = -d.y / d.z; tr.Sz = 1.f / d.z;
    return tr;
}

PD bool tri_hit_pre(const f3 &ro, const TriRay &tr, float tmax, const f3 &p0, const f3 &p1, const f3 &p2, float *tHit, float *ob0,
                    float *ob1, float *ob2) {
    f3 p0t = p0 - ro, p1t = p1 - ro, p2t = p2 - ro;
    p0t = f3(comp(p0t, tr.kx), comp(p0t, tr.ky), comp(p0t, tr.kz));
    p1t = f3(comp(p1t, tr.kx), comp(p1t, tr.ky), comp(p1t, tr.kz));
    p2t = f3(comp(p2t, tr.kx), comp(p2t, tr.ky), comp(p2t, tr.kz));
    const float Sx = tr.Sx, Sy = tr.Sy, Sz = tr.Sz;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (is_zero(e0) || is_zero(e1) || is_zero(e2)) {
        double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    float det = e0 + e1 + e2;
    if (is_zero(det)) return false;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < tmax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > tmax * det)) return false;
    float invDet = 1 / det;
    float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    float t = tScaled * invDet;
    float maxZt = max_comp(abs3(f3(p0t.z, p1t.z, p2t.z)));
    float deltaZ = gammar(3) * maxZt;
    float maxXt = max_comp(abs3(f3(p0t.x, p1t.x, p2t.x)));
    float maxYt = max_comp(abs3(f3(p0t.y, p1t.y, p2t.y)));
    float deltaX = gammar(5) * (maxXt + maxZt);
    float deltaY = gammar(5) * (maxYt + maxZt);
    float deltaE = 2 * (gammar(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = max_comp(abs3(f3(e0, e1, e2)));
    float deltaT = 3 * (gammar(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * absr(invDet);
    if (t <= deltaT) return false;
    *tHit = t;
    *ob0 = b0; *ob1 = b1; *ob2 = b2;
    return true;
}
/* The same test cut in two for the tracer's pair mode (dev_trace.h): everything that does not depend on tMax here - by whichever
 * lane has been dealt this (ray, triangle) pair - and the two comparisons against tMax * det (mesh.cpp:216-219) in tri_cand_accept,
 * made by the ray's own lane with the tMax it has when the triangle's turn comes.  The same operations on the same values as
 * tri_hit_pre.  No candidate: *det = 0 (tri_hit_pre returns false on a zero determinant before it looks at tMax). */
PD void tri_cand(const f3 &ro, const TriRay &tr, const f3 &p0, const f3 &p1, const f3 &p2, float *tHit, float *otScaled, float *odet, float *ob0,
                 float *ob1) {
    *odet = 0.f;
    f3 p0t = p0 - ro, p1t = p1 - ro, p2t = p2 - ro;
    p0t = f3(comp(p0t, tr.kx), comp(p0t, tr.ky), comp(p0t, tr.kz));
    p1t = f3(comp(p1t, tr.kx), comp(p1t, tr.ky), comp(p1t, tr.kz));
    p2t = f3(comp(p2t, tr.kx), comp(p2t, tr.ky), comp(p2t, tr.kz));
    const float Sx = tr.Sx, Sy = tr.Sy, Sz = tr.Sz;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (is_zero(e0) || is_zero(e1) || is_zero(e2)) {
        double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return;
    float det = e0 + e1 + e2;
    if (is_zero(det)) return;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && tScaled >= 0) return;
    else if (det > 0 && tScaled <= 0) return;
    float invDet = 1 / det;
    float b0 = e0 * invDet, b1 = e1 * invDet;
    float t = tScaled * invDet;
    float maxZt = max_comp(abs3(f3(p0t.z, p1t.z, p2t.z)));
    float deltaZ = gammar(3) * maxZt;
    float maxXt = max_comp(abs3(f3(p0t.x, p1t.x, p2t.x)));
    float maxYt = max_comp(abs3(f3(p0t.y, p1t.y, p2t.y)));
    float deltaX = gammar(5) * (maxXt + maxZt);
    float deltaY = gammar(5) * (maxYt + maxZt);
    float deltaE = 2 * (gammar(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = max_comp(abs3(f3(e0, e1, e2)));
    float deltaT = 3 * (gammar(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * absr(invDet);
    if (t <= deltaT) return;
    *tHit = t;
    *otScaled = tScaled;
    *odet = det;
    *ob0 = b0; *ob1 = b1;
}
PD bool tri_cand_accept(float tScaled, float det, float tmax) {
    if (is_zero(det)) return false;
    if (det < 0 && tScaled < tmax * det) return false;
    else if (det > 0 && tScaled > tmax * det) return false;
    return true;
}
PD bool tri_hit(const f3 &ro, const f3 &rd, float tmax, const f3 &p0, const f3 &p1, const f3 &p2, float *tHit, float *ob0,
                float *ob1, float *ob2) {
    TriRay tr = tri_ray_setup(rd);
    return tri_hit_pre(ro, tr, tmax, p0, p1, p2, tHit, ob0, ob1, ob2);
}

/* ---- sphere, sphere.cpp:105-127.  Object-space ray given.  Every sphere the scene API can make
 * is a full sphere (shape.h:74-78), for which the clipping branch of sphere.cpp:136-150 is never
 * taken, so only t selection is needed here. */
PD bool sphere_hit_t_obj(const f3 &o, const f3 &d, float radius, float tmax, float *tHit) {
    float a = d.x * d.x + d.y * d.y + d.z * d.z;
    float b = 2 * (d.x * o.x + d.y * o.y + d.z * o.z);
    float c = o.x * o.x + o.y * o.y + o.z * o.z - radius * radius;
    float t0, t1;
    if (!quadratic(a, b, c, &t0, &t1)) return false;
    if (t0 > tmax || t1 <= 0) return false;
    float ts = t0;
    if (ts <= 0 || is_unsafe_hit(ts)) {
        ts = t1;
        if (ts > tmax) return false;
    }
    *tHit = ts;
    return true;
}
/* particle: toObject = Inverse(Translate(c)) (particlecloud.cpp:113-115).  With a pure
 * translation xf_point_err reduces exactly to o - c with error gamma(3)*(|o|+|c|) and the
 * direction is unchanged, so no matrices are built. */
PD void particle_object_ray(const f3 &ro, const f3 &rd, const f3 &c, f3 *o, f3 *d) {
    f3 oo = ro - c; /* (1*x + 0*y) + (0*z + (-c)) */
    f3 oErr = gammar(3) * f3(absr(ro.x) + absr(-c.x), absr(ro.y) + absr(-c.y), absr(ro.z) + absr(-c.z));
    float l2 = length2(rd);
    if (l2 > 0) {
        float dt = dot(abs3(rd), oErr) / l2;
        oo += rd * dt;
    }
    *o = oo;
    *d = rd;
}

/* The reference reaches IntersectSphere for a particle only through the leaf that holds it, and its median-split tree
 * (BVH_MAX_DEPTH 20, primitive.cpp:8) gives every particle a leaf of its own up to 2^20 particles per cloud: the
 * particle's own box (ParticleCloudGetBounds, particlecloud.cpp:284-293: +-radius about the centre) is tested first
 * (particlecloud.cpp:226-227).  The float quadratic accepts some grazing rays that miss the exact sphere - and its box;
 * the reference never sees those hits, our wider leaves (up to four particles) would.  Found at 4 M particles: two of
 * 85,510 primary rays. */
PD bool particle_box_test(const f3 &ro, const f3 &inv, float tmax, const f3 &c, float radius) {
    float te;
    return slab_test6(-radius + c.x, -radius + c.y, -radius + c.z, radius + c.x, radius + c.y, radius + c.z, ro, inv, tmax, &te);
}

/* ---- rectangle, rectangle.cpp:3-21 ---- */
PD bool rect_hit_t(const DPrim &pr, const f3 &ro, const f3 &rd, float tmax, float *tHit) {
    f3 o, d;
    xf_ray_err(pr.w2o, ro, rd, &o, &d);
    float hx = pr.p[0] / 2, hy = pr.p[1] / 2;
    float zb = kShadowEpsilon * fmaxr(hx, hy);
    float bmin[3] = {-hx, -hy, -zb}, bmax[3] = {hx, hy, zb};
    f3 inv(1 / d.x, 1 / d.y, 1 / d.z);
    float tmin;
    if (!slab_test(bmin, bmax, o, inv, tmax, &tmin)) return false;
    if (is_unsafe_hit(tmin)) return false;
    f3 pHit = o + d * tmin;
    if (pHit.x < -hx || pHit.x > hx || pHit.y < -hy || pHit.y > hy) return false;
    *tHit = tmin;
    return true;
}
/* ---- disk, disk.cpp:8-25 ---- */
PD bool disk_hit_t(const DPrim &pr, const f3 &ro, const f3 &rd, float tmax, float *tHit) {
    f3 o, d;
    xf_ray_err(pr.w2o, ro, rd, &o, &d);
    float height = pr.p[0], radius = pr.p[1], inner = pr.p[2], phiMax = pr.p[3];
    if (is_zero(d.z)) return false;
    float ts = (height - o.z) / d.z;
    if (ts <= 0 || ts >= tmax) return false;
    f3 pHit = o + d * ts;
    float dist2v = pHit.x * pHit.x + pHit.y * pHit.y;
    if (dist2v > radius * radius || dist2v < inner * inner) return false;
    float phi = atan2f(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * kPi;
    if (phi > phiMax) return false;
    *tHit = ts;
    return true;
}

/* ---- procedural (SDF) shapes, procedural.cpp ---- */
PD float sdf_box(const f3 &p, const f3 &b) { /* :78-81 */
    f3 d = abs3(p) - b;
    return fminr(fmaxr(d.x, fmaxr(d.y, d.z)), 0.0f) + length(f3(fmaxr(d.x, 0.f), fmaxr(d.y, 0.f), fmaxr(d.z, 0.f)));
}
/* SDF::Triangle (procedural.cpp:112-130) against a record whose point-independent parts (edges,
 * normal, the three double-precision cross products, squared lengths) were made on the host with the
 * reference's arithmetic (ptr_toys.cpp) */
PD float sdf_sign(float x) { /* procedural.cpp:14-18 */
    if (is_zero(x)) return 0;
    return x > 0 ? 1.f : -1.f;
}
PD float sdf_toy_triangle(const DToyTri &t, const f3 &p) {
    f3 pa = p - f3(t.a[0], t.a[1], t.a[2]), pb = p - f3(t.b[0], t.b[1], t.b[2]), pc = p - f3(t.c[0], t.c[1], t.c[2]);
    f3 ba(t.ba[0], t.ba[1], t.ba[2]), cb(t.cb[0], t.cb[1], t.cb[2]), ac(t.ac[0], t.ac[1], t.ac[2]);
    float ss = sdf_sign(dot(f3(t.e0[0], t.e0[1], t.e0[2]), pa)) + sdf_sign(dot(f3(t.e1[0], t.e1[1], t.e1[2]), pb)) +
               sdf_sign(dot(f3(t.e2[0], t.e2[1], t.e2[2]), pc));
    if (ss < 2) {
        float a2 = length2(ba * clampr(dot(ba, pa) / t.d2ba, 0.f, 1.f) - pa);
        float b2 = length2(cb * clampr(dot(cb, pb) / t.d2cb, 0.f, 1.f) - pb);
        float c2 = length2(ac * clampr(dot(ac, pc) / t.d2ac, 0.f, 1.f) - pc);
        return sqrtf(fminr(a2, fminr(b2, c2)));
    }
    float dn = dot(f3(t.nor[0], t.nor[1], t.nor[2]), pa);
    return sqrtf(dn * dn / t.d2nor);
}
PD float proc_map(const DScene &sc, const DPrim &pr, const f3 &p, int *shapeId) {
    if (pr.type == SH_SPHERE_PROC) return length(p) - pr.p[0];
    if (pr.type == SH_BOX_PROC) return sdf_box(p, f3(pr.p[0], pr.p[1], pr.p[2]) * 0.5f);
    if (pr.toy_id >= 0 && pr.toy_id <= 3) {
        /* origami toys, procedural.cpp:402-574: mirrored in z (the boat, id 0, also in x); the nearest
         * triangle's id becomes the shape id (first one on ties), the dragon (id 1) always reports 5 */
        f3 q(pr.toy_id == 0 ? absr(p.x) : p.x, p.y, absr(p.z));
        float best = kInfinity;
        int id = -1;
        for (int i = 0; i < pr.geom; i++) {
            const DToyTri &t = sc.toy_tris[pr.root + i];
            float d = sdf_toy_triangle(t, q);
            if (best > d) { best = d; id = (int)t.id; }
        }
        *shapeId = pr.toy_id == 1 ? 5 : id;
        return best;
    }
    if (pr.toy_id != 4) return 0; /* "Unknown component id": Map returns false (:389-394) */
    /* SH_TOY id 4 "Terrain" (procedural.cpp:380-400): box over the component's extents */
    f3 len(absr(pr.p[3] - pr.p[0]), absr(pr.p[4] - pr.p[1]), absr(pr.p[5] - pr.p[2]));
    return sdf_box(p, len * 0.5f);
}
PD void proc_local_bounds(const DPrim &pr, float *bmin, float *bmax) {
    if (pr.type == SH_SPHERE_PROC) {
        float r = pr.p[0];
        bmin[0] = bmin[1] = bmin[2] = -r; bmax[0] = bmax[1] = bmax[2] = r;
    } else if (pr.type == SH_BOX_PROC) {
        f3 h = f3(pr.p[0], pr.p[1], pr.p[2]) * 0.5f;
        bmin[0] = -h.x; bmin[1] = -h.y; bmin[2] = -h.z; bmax[0] = h.x; bmax[1] = h.y; bmax[2] = h.z;
    } else {
        for (int i = 0; i < 3; i++) { bmin[i] = pr.p[i]; bmax[i] = pr.p[3 + i]; }
    }
}
PD bool inside_box(const f3 &p, const float *bmin, const float *bmax) {
    return p.x >= bmin[0] && p.x <= bmax[0] && p.y >= bmin[1] && p.y <= bmax[1] && p.z >= bmin[2] && p.z <= bmax[2];
}
/* :187-227.  tHit is the distance along the NORMALISED direction (sic, :246). */
PD bool proc_hit_t(const DScene &sc, const DPrim &pr, const f3 &ro, const f3 &rd, float rtmax, float *tHit, int *shapeId) {
    f3 un = normalize(rd);
    f3 o, d;
    xf_ray_err(pr.w2o, ro, un, &o, &d);
    float tMax = rtmax * length(rd);
    float bmin[3], bmax[3];
    proc_local_bounds(pr, bmin, bmax);
    float t = 0;
    if (!inside_box(o, bmin, bmax)) {
        f3 inv(1 / d.x, 1 / d.y, 1 / d.z);
        float t1, t2;
        if (!slab_test(bmin, bmax, o, inv, kInfinity, &t1, &t2)) return false;
        t = fminr(t1, t2);
    }
    int id = 0;
    bool hit = false;
    for (int i = 0; i < 256 && t < tMax; i++) {
        f3 p = o + d * t;
        float dd = proc_map(sc, pr, p, &id);
        if ((double)absr(dd) < (0.001 * t)) { hit = true; break; }
        t += dd;
        if (!inside_box(o + d * t, bmin, bmax)) break;
    }
    if (hit) { *tHit = t; *shapeId = id; }
    return hit;
}

/* Shape::Intersect of an analytic primitive: hit distance only (the interaction is rebuilt by
 * make_si for the one hit that survives) */
/* PROC = false: for scenes without SDF shapes (DScene::has_proc) - their sphere tracing is what a tracer kernel's registers are
 * sized by (70-80 spilled at 96 with it, none without) */
template <bool PROC = true>
PD bool shape_hit_t(const DScene &sc, const DPrim &pr, const f3 &ro, const f3 &rd, float tmax, float *tHit, int *face) {
    *face = 0;
    switch (pr.type) {
    case SH_SPHERE: {
        f3 o, d;
        xf_ray_err(pr.w2o, ro, rd, &o, &d);
        return sphere_hit_t_obj(o, d, pr.p[0], tmax, tHit);
    }
    case SH_RECT: return rect_hit_t(pr, ro, rd, tmax, tHit);
    case SH_DISK: return disk_hit_t(pr, ro, rd, tmax, tHit);
    case SH_SPHERE_PROC: case SH_BOX_PROC: case SH_TOY:
        if (PROC) return proc_hit_t(sc, pr, ro, rd, tmax, tHit, face);
        return false;
    default: return false;
    }
}

/* ---- traversal -------------------------------------------------------------------------------
 * Two levels, as in the reference, but laid out differently:
 *
 *  top   the reference's own median-split BVH over primitives (tens of nodes), walked exactly as
 *        Aggregator::Intersect walks it (primitive.cpp:374-431): a few answers depend on the
 *        order in which primitives are met (an SDF hit distance is measured along the normalised
 *        direction and can inflate tMax of a non-unit shadow ray, procedural.cpp:238; a
 *        rectangle's hit can lie in front of its bounds, rectangle.cpp:8 vs :37-40).
 *  geom  meshes and particle clouds: one SAH-built BVH each, collapsed to 4-wide 128-byte nodes
 *        (the four children's boxes in one line), 48-byte pre-gathered leaf records, children
 *        visited nearest first (replaces the nested loops of mesh.cpp:454-509 and
 *        particlecloud.cpp:197-274).
 *
 * Closest hit: the reference accepts t <= tMax, so of two hits at bit-equal t the later tested
 * wins; inside a mesh that depends on its BVH - the oracle reports such rays and the tests
 * exclude them.  ANY = true returns at the first accepted hit. */
struct TraceStats { unsigned nodes, items; };

/* Per-lane traversal stack: the first LDS_STACK entries live in LDS (column threadIdx.x of a
 * [LDS_STACK][block] array: conflict-free, and a pop no longer waits on a scratch load), deeper
 * entries spill to a small scratch array that well-built trees rarely reach. */
#ifndef LDS_STACK
#define LDS_STACK 24
#endif
#define OVF_STACK (BVH_STACK - LDS_STACK)
#define TRACE_TPB 256
struct LaneStack {
    int *lds; /* &smem[threadIdx.x], stride TRACE_TPB */
};
#define STACK_PUSH(v)                                   \
    do {                                                \
        if (sp < LDS_STACK) ls.lds[sp * TRACE_TPB] = (v); \
        else if (sp < BVH_STACK) ovf[sp - LDS_STACK] = (v); \
        sp++;                                           \
    } while (0)
#define STACK_POP() (--sp, sp < LDS_STACK ? ls.lds[sp * TRACE_TPB] : ovf[(sp < BVH_STACK ? sp : BVH_STACK - 1) - LDS_STACK])

template <bool ANY, bool COUNT>
PD bool bvh_closest(const DScene &sc, int root, const f3 &ro, const f3 &rd, const f3 &inv, float &tmax, int &hface, float &hb0,
                    float &hb1, TraceStats *st, const LaneStack &ls) {
    int ovf[OVF_STACK];
    int sp = 0;
    bool hit = false;
    int node = root;
    const TriRay tr = tri_ray_setup(rd);
    const int DONE = 0x7FFFFFFF;
    while (true) {
        /* descend until this lane holds a leaf (or has nothing left) */
        while (node >= 0 && node != DONE) {
            int r0, r1, r2, r3;
            const int nh = bvh4_visit(sc.nodes, node, ro, inv, tmax, r0, r1, r2, r3, COUNT ? &st->nodes : nullptr);
            if (nh == 0) node = sp == 0 ? DONE : STACK_POP();
            else {
                if (nh > 3) STACK_PUSH(r3);
                if (nh > 2) STACK_PUSH(r2);
                if (nh > 1) STACK_PUSH(r1);
                node = r0;
            }
        }
        if (node == DONE) break;
        {
            unsigned packed = (unsigned)(-(node + 1));
            unsigned first = packed & 0x07FFFFFFu, n = (packed >> 27) + 1;
            for (unsigned i = 0; i < n; i++) {
                const float4 *ip = reinterpret_cast<const float4 *>(sc.items + first + i);
                float4 q0 = ip[0], q1 = ip[1], q2 = ip[2];
                int kind = __float_as_int(q2.w);
                if (COUNT) st->items += 1;
                if (kind == ITEM_TRI) {
                    float t, b0, b1, b2;
                    if (tri_hit_pre(ro, tr, tmax, f3(q0.x, q0.y, q0.z), f3(q1.x, q1.y, q1.z), f3(q2.x, q2.y, q2.z), &t, &b0, &b1, &b2)) {
                        tmax = t; hface = __float_as_int(q1.w); hb0 = b0; hb1 = b1; hit = true;
                        if (ANY) return true;
                    }
                } else {
                    f3 o, d;
                    particle_object_ray(ro, rd, f3(q0.x, q0.y, q0.z), &o, &d);
                    float t;
                    if (particle_box_test(ro, inv, tmax, f3(q0.x, q0.y, q0.z), q1.x) && sphere_hit_t_obj(o, d, q1.x, tmax, &t)) {
                        tmax = t; hface = __float_as_int(q1.w); hb0 = 0; hb1 = 0; hit = true;
                        if (ANY) return true;
                    }
                }
            }
        }
        if (sp == 0) break;
        node = STACK_POP();
    }
    return hit;
}

/* PrimitiveIntersect (primitive.cpp:52-73) for the primitive of a top-level leaf */
template <bool ANY, bool COUNT>
PD bool prim_closest(const DScene &sc, int p, const f3 &ro, const f3 &rd, const f3 &inv, float &tmax, int &hprim, int &hface,
                     float &hb0, float &hb1, TraceStats *st, const LaneStack &ls) {
    const DPrim &pr = sc.prims[p];
    if (pr.type == SH_MESH || pr.type == SH_CLOUD) {
        /* Mesh::Intersect / ParticleCloud::Intersect start with their own root box, which is
         * the primitive's bound the leaf just passed: go straight to the geometry BVH */
        if (bvh_closest<ANY, COUNT>(sc, pr.root, ro, rd, inv, tmax, hface, hb0, hb1, st, ls)) { hprim = p; return true; }
        return false;
    }
    float t;
    int face;
    if (COUNT) st->items += 1;
    if (shape_hit_t(sc, pr, ro, rd, tmax, &t, &face)) {
        tmax = t; hprim = p; hface = face; hb0 = 0; hb1 = 0;
        return true;
    }
    return false;
}

/* Aggregator::Intersect, primitive.cpp:374-431, step for step over the reference's own tree */
template <bool ANY, bool COUNT>
PD bool traverse(const DScene &sc, const f3 &ro, const f3 &rd, float &tmax, int &hprim, int &hface, float &hb0, float &hb1,
                 TraceStats *st, const LaneStack &ls) {
    if (sc.n_top == 0) return false;
    f3 inv(1 / rd.x, 1 / rd.y, 1 / rd.z);
    bool hit = false;
    int stack[TOP_STACK];
    int sp = 0;
    stack[sp++] = -1;
    int node = 0;
    while (node != -1) {
        const TopNode &nd = sc.top[node];
        float te;
        if (COUNT) st->nodes += 1;
        if (slab_test(nd.bmin, nd.bmax, ro, inv, tmax, &te)) {
            if (nd.left < 0) {
                for (int i = 0; i < nd.n; i++) {
                    if (prim_closest<ANY, COUNT>(sc, sc.order[nd.first + i], ro, rd, inv, tmax, hprim, hface, hb0, hb1, st, ls)) {
                        hit = true;
                        if (ANY) return true;
                    }
                }
                node = stack[--sp];
            } else {
                const TopNode &l = sc.top[nd.left];
                const TopNode &r = sc.top[nd.right];
                float tl0, tr0;
                if (COUNT) st->nodes += 2;
                bool visitL = slab_test(l.bmin, l.bmax, ro, inv, tmax, &tl0);
                bool visitR = slab_test(r.bmin, r.bmax, ro, inv, tmax, &tr0);
                if (visitR && visitL) {
                    int first, second;
                    if (tr0 < tl0) { first = nd.right; second = nd.left; }
                    else { first = nd.left; second = nd.right; }
                    if (sp < TOP_STACK) stack[sp++] = second;
                    node = first;
                } else if (visitL) node = nd.left;
                else if (visitR) node = nd.right;
                else node = stack[--sp];
            }
        } else node = stack[--sp];
    }
    return hit;
}

/* ---- SurfaceInteraction reconstruction ------------------------------------------------------- */
/* SurfaceInteraction ctor (interaction.cpp:23-39) + Transform::operator()(SurfaceInteraction)
 * (transform.cpp:224-256; note pError is NOT carried through the transform) for the
 * object-space shapes. */
PD void si_from_object(const DPrim &pr, const float *o2w, const float *w2o, const f3 &pHit, const f2 &uv, const f3 &wo_obj,
                       const f3 &dpdu, const f3 &dpdv, int face, SI *si) {
    f3 n = normalize(cross(dpdu, dpdv));
    f3 wo = normalize(wo_obj);
    if (pr.flags & PF_FLIP_N) n *= -1.0f;
    si->p = xf_point(o2w, pHit);
    si->n = normalize(xf_normal(w2o, n));
    si->wo = normalize(xf_vec(o2w, wo));
    si->pError = f3(0.f);
    si->uv = uv;
    si->dpdu = xf_vec(o2w, dpdu);
    si->face = face;
}

/* sphere.cpp:129-190 / particlecloud.cpp:133-175 given the accepted t */
PD void sphere_si(const DPrim &pr, const float *o2w, const float *w2o, const f3 &o, const f3 &d, float t, float radius, bool cloud,
                  int face, SI *si) {
    f3 pHit = o + d * t;
    pHit *= radius / dist(pHit, f3(0, 0, 0));
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;
    float phi = atan2f(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * kPi;
    float theta = acosf(clampr(pHit.z / radius, -1, 1));
    float zRadius = sqrtf(pHit.x * pHit.x + pHit.y * pHit.y);
    float invZRadius = 1 / zRadius;
    float cosPhi = pHit.x * invZRadius, sinPhi = pHit.y * invZRadius;
    float u, v;
    f3 dpdu, dpdv;
    if (!cloud) {
        float thetaMin = pr.p[3], thetaMax = pr.p[4], phiMax = pr.p[5];
        u = phi / phiMax;
        v = (theta - thetaMin) / (thetaMax - thetaMin);
        dpdu = f3(-phiMax * pHit.y, phiMax * pHit.x, 0);
        dpdv = (thetaMax - thetaMin) * f3(pHit.z * cosPhi, pHit.z * sinPhi, -radius * sinf(theta));
    } else {
        float phiMax = 2 * kPi;
        u = phi / (2 * kPi);
        v = 1.0 - theta / kPi;
        dpdu = f3(-phiMax * pHit.y, phiMax * pHit.x, 0);
        dpdv = (float)(-kPi) * f3(pHit.z * cosPhi, pHit.z * sinPhi, -radius * sinf(theta));
    }
    si_from_object(pr, o2w, w2o, pHit, f2(u, v), -d, dpdu, dpdv, face, si);
}

/* procedural.cpp:145-185,228-247,263-352 */
PD f3 proc_derivative_normal(const DScene &sc, const DPrim &pr, const f3 &p) {
    bool found = false;
    float scale = 0.001;
    f3 n;
    int it = 0, id;
    while (!found) {
        float e = scale * length(p);
        float x = proc_map(sc, pr, f3(p.x + e, p.y, p.z), &id) - proc_map(sc, pr, f3(p.x - e, p.y, p.z), &id);
        float y = proc_map(sc, pr, f3(p.x, p.y + e, p.z), &id) - proc_map(sc, pr, f3(p.x, p.y - e, p.z), &id);
        float z = proc_map(sc, pr, f3(p.x, p.y, p.z + e), &id) - proc_map(sc, pr, f3(p.x, p.y, p.z - e), &id);
        n = f3(x, y, z);
        found = !is_zero(length(n));
        if (!found) {
            if (it++ > 10) { n = f3(0, 1, 0); break; }
            scale *= 2;
        }
    }
    return normalize(n);
}
PD void proc_si(const DScene &sc, const DPrim &pr, const f3 &ro, const f3 &rd, float t, int face, SI *si) {
    f3 un = normalize(rd);
    f3 o, d;
    xf_ray_err(pr.w2o, ro, un, &o, &d);
    f3 pHit = o + d * t;
    f2 uv(0, 0);
    f3 n;
    bool refined = false;
    if (pr.type == SH_SPHERE_PROC) {
        float radius = pr.p[0];
        face = 0;
        pHit *= radius / dist(pHit, f3(0, 0, 0));
        float phi = atan2f(pHit.y, pHit.x);
        if (phi < 0) phi += 2 * kPi;
        float u = phi / (2 * kPi);
        float theta = acosf(clampr(pHit.z / radius, -1, 1));
        float v = (kPi - theta) / kPi;
        uv = f2(u, v);
        n = f3(pHit.x, pHit.y, pHit.z) / radius;
        refined = true;
    } else if (pr.type == SH_BOX_PROC) {
        f3 len(pr.p[0], pr.p[1], pr.p[2]);
        f3 half = len * 0.5f;
        float cx = kInfinity, cy = -1, cz = -1;
        float q[6][3] = {{absr(pHit.x - half.x), 1, +half.x}, {absr(pHit.x + half.x), 3, -half.x}, {absr(pHit.y - half.y), 4, +half.y},
                         {absr(pHit.y + half.y), 5, -half.y}, {absr(pHit.z - half.z), 2, +half.z}, {absr(pHit.z + half.z), 0, -half.z}};
        for (int i = 0; i < 6; i++)
            if (cx > q[i][0]) { cx = q[i][0]; cy = q[i][1]; cz = q[i][2]; }
        int target = (int)cy;
        float u = 0, v = 0;
        if (target == 1 || target == 3) {
            pHit.x = cz;
            n = cz > 0 ? f3(1, 0, 0) : f3(-1, 0, 0);
            u = (pHit.z + half.z) / len.z;
            v = 1.f - (pHit.y + half.y) / len.y;
        } else if (target == 4 || target == 5) {
            pHit.y = cz;
            n = cz > 0 ? f3(0, 1, 0) : f3(0, -1, 0);
            u = (pHit.x + half.x) / len.x;
            v = (pHit.z + half.z) / len.z;
        } else {
            pHit.z = cz;
            n = cz > 0 ? f3(0, 0, 1) : f3(0, 0, -1);
            u = (pHit.x + half.x) / len.x;
            v = 1.f - (pHit.y + half.y) / len.z; /* sic */
        }
        uv = f2(u, v);
        refined = true;
    }
    if (!refined) n = proc_derivative_normal(sc, pr, pHit);
    f3 dpdu, dpdv;
    coordinate_system(n, &dpdu, &dpdv);
    si_from_object(pr, pr.o2w, pr.w2o, pHit, uv, -d, dpdu, dpdv, face, si);
}

/* mesh.cpp:71-90 */
PD void mesh_uvs(const DScene &sc, const DPrim &pr, int tri, f2 st[3]) {
    const DMesh &m = sc.meshes[pr.geom];
    st[0] = f2(0, 0); st[1] = f2(1, 0); st[2] = f2(1, 1);
    if (m.uv) {
        const int32_t *ix = sc.uvidx + 3 * (m.tri_base + tri);
        int i0 = ix[0], i1 = ix[1], i2 = ix[2];
        if (!((i0 < 0 || i0 > m.n_uvs) || (i1 < 0 || i1 > m.n_uvs) || (i2 < 0 || i2 > m.n_uvs))) {
            st[0] = f2(m.uv[2 * i0], 1.0 - m.uv[2 * i0 + 1]);
            st[1] = f2(m.uv[2 * i1], 1.0 - m.uv[2 * i1 + 1]);
            st[2] = f2(m.uv[2 * i2], 1.0 - m.uv[2 * i2 + 1]);
        }
    }
}
PD void tri_vertices(const DScene &sc, const DPrim &pr, int tri, f3 *p0, f3 *p1, f3 *p2) {
    const float *v = sc.tris + 9 * (sc.meshes[pr.geom].tri_base + tri);
    *p0 = f3(v[0], v[1], v[2]); *p1 = f3(v[3], v[4], v[5]); *p2 = f3(v[6], v[7], v[8]);
}
/* mesh.cpp:251-303 */
PD bool tri_si(const DScene &sc, const DPrim &pr, const f3 &ro, const f3 &rd, int tri, SI *si) {
    f3 p0, p1, p2;
    tri_vertices(sc, pr, tri, &p0, &p1, &p2);
    float t, b0, b1, b2;
    if (!tri_hit(ro, rd, kInfinity, p0, p1, p2, &t, &b0, &b1, &b2)) return false;
    f2 st[3];
    mesh_uvs(sc, pr, tri, st);
    f2 dst02 = st[0] - st[2], dst12 = st[1] - st[2];
    f3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = dst02.x * dst12.y - dst02.y * dst12.x;
    bool degenerateUV = (double)absr(determinant) < 1e-8;
    if (degenerateUV) {
        st[0] = f2(0, 0); st[1] = f2(1, 0); st[2] = f2(1, 1);
        dst02 = st[0] - st[2]; dst12 = st[1] - st[2];
        determinant = dst02.x * dst12.y - dst02.y * dst12.x;
        degenerateUV = (double)absr(determinant) < 1e-8;
    }
    f3 dpdu;
    if (!degenerateUV) {
        float invdet = 1 / determinant;
        dpdu = (dst12.y * dp02 - dst02.y * dp12) * invdet;
    }
    float xAbsSum = (absr(b0 * p0.x) + absr(b1 * p1.x) + absr(b2 * p2.x));
    float yAbsSum = (absr(b0 * p0.y) + absr(b1 * p1.y) + absr(b2 * p2.y));
    float zAbsSum = (absr(b0 * p0.z) + absr(b1 * p1.z) + absr(b2 * p2.z));
    si->pError = gammar(7) * f3(xAbsSum, yAbsSum, zAbsSum);
    si->p = b0 * p0 + b1 * p1 + b2 * p2;
    si->uv = b0 * st[0] + b1 * st[1] + b2 * st[2];
    si->wo = normalize(-rd);
    si->dpdu = dpdu;
    si->n = normalize(cross(dp02, dp12));
    if (pr.flags & PF_FLIP_N) si->n = -si->n;
    si->face = tri;
    return true;
}

/* Rebuild the interaction of an accepted hit (prim, face, t) on ray (ro, rd); sets medium
 * interface as PrimitiveIntersect does (primitive.cpp:66-71). */
PD void make_si(const DScene &sc, int prim, int face, const f3 &ro, const f3 &rd, float t, int ray_medium, SI *si) {
    const DPrim &pr = sc.prims[prim];
    switch (pr.type) {
    case SH_SPHERE: {
        f3 o, d;
        xf_ray_err(pr.w2o, ro, rd, &o, &d);
        sphere_si(pr, pr.o2w, pr.w2o, o, d, t, pr.p[0], false, 0, si);
    } break;
    case SH_CLOUD: {
        const DCloud &c = sc.clouds[pr.geom];
        f3 ctr(c.pos[3 * face], c.pos[3 * face + 1], c.pos[3 * face + 2]);
        f3 o, d;
        particle_object_ray(ro, rd, ctr, &o, &d);
        float o2w[12] = {1, 0, 0, ctr.x, 0, 1, 0, ctr.y, 0, 0, 1, ctr.z};
        float w2o[12] = {1, 0, 0, -ctr.x, 0, 1, 0, -ctr.y, 0, 0, 1, -ctr.z};
        sphere_si(pr, o2w, w2o, o, d, t, c.radius, true, face, si);
    } break;
    case SH_RECT: { /* rectangle.cpp:19-33 */
        f3 o, d;
        xf_ray_err(pr.w2o, ro, rd, &o, &d);
        float hx = pr.p[0] / 2, hy = pr.p[1] / 2;
        f3 pHit = o + d * t;
        float u = absr((pHit.x - hx) / pr.p[0]);
        float v = absr((pHit.y - hy) / pr.p[1]);
        si_from_object(pr, pr.o2w, pr.w2o, pHit, f2(u, v), -d, f3(1, 0, 0), f3(0, 1, 0), 0, si);
    } break;
    case SH_DISK: { /* disk.cpp:18-40 */
        f3 o, d;
        xf_ray_err(pr.w2o, ro, rd, &o, &d);
        float height = pr.p[0], radius = pr.p[1], inner = pr.p[2], phiMax = pr.p[3];
        f3 pHit = o + d * t;
        float dist2v = pHit.x * pHit.x + pHit.y * pHit.y;
        float phi = atan2f(pHit.y, pHit.x);
        if (phi < 0) phi += 2 * kPi;
        float u = phi / phiMax;
        float rHit = sqrtf(dist2v);
        float v = (radius - rHit) / (radius - inner);
        f3 dpdu(-phiMax * pHit.y, phiMax * pHit.x, 0);
        f3 dpdv = f3(pHit.x, pHit.y, 0.) * (inner - radius) / rHit;
        pHit.z = height;
        si_from_object(pr, pr.o2w, pr.w2o, pHit, f2(u, v), -d, dpdu, dpdv, 0, si);
    } break;
    case SH_MESH: tri_si(sc, pr, ro, rd, face, si); break;
    default: proc_si(sc, pr, ro, rd, t, face, si); break;
    }
    si->prim = prim;
    if (pr.med_inside != pr.med_outside) { si->med_in = pr.med_inside; si->med_out = pr.med_outside; }
    else { si->med_in = ray_medium; si->med_out = ray_medium; }
}

/* Interaction::SpawnRay / SpawnRayTo, interaction.cpp:5-21 */
PD int si_medium(const f3 &n, int med_in, int med_out, const f3 &w) { return dot(w, n) > 0 ? med_out : med_in; }
