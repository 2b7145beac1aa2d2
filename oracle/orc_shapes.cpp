/*
 * orc_shapes.cpp — ORACLE (test infrastructure only).  PARITY UNPINNED.
 * Restates: BVH build (src/core/primitive.cpp:507-618, util.h:10-39), shape intersection /
 * sampling (src/shapes/*.cpp, src/core/procedural.cpp), aggregate traversal
 * (src/core/primitive.cpp:52-73,360-461), ray spawning (src/core/interaction.cpp:5-39).
 */
#include "orc_scene.h"
#include <cstdio>

namespace orc {

/* ---- glibc rand(), TYPE_3 ------------------------------------------------------------ */
GlibcRand::GlibcRand(uint32_t seed) {
    int32_t word = seed ? (int32_t)seed : 1;
    buf[0] = (uint32_t)word;
    for (int i = 1; i < 31; i++) {
        int32_t hi = word / 127773, lo = word % 127773;
        word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        buf[i] = (uint32_t)word;
    }
    f = 3;
    b = 0;
    for (int i = 0; i < 310; i++) next();
}
int GlibcRand::next() {
    buf[f] += buf[b];
    int result = (int)(buf[f] >> 1);
    f = (f + 1) % 31;
    b = (b + 1) % 31;
    return result;
}

/* ---- util.h:10-39 iterative quicksort, compare(a,b) = a.pMin[axis] >= b.pMin[axis] ------- */
static inline bool cmp_ge(const PrimitiveHandle &a, const PrimitiveHandle &b, int axis) {
    return a.bound.pMin[axis] >= b.bound.pMin[axis];
}
static bool QuickSortHandles(PrimitiveHandle *arr, int elements, int axis) {
    const int MAX_LEVELS = 1000;
    int beg[MAX_LEVELS], end[MAX_LEVELS], i = 0, L, R;
    PrimitiveHandle piv;
    beg[0] = 0;
    end[0] = elements;
    while (i >= 0) {
        L = beg[i];
        R = end[i] - 1;
        if (L < R) {
            piv = arr[L];
            if (i == MAX_LEVELS - 1) return false;
            while (L < R) {
                while (cmp_ge(arr[R], piv, axis) && L < R) R--;
                if (L < R) arr[L++] = arr[R];
                while (cmp_ge(piv, arr[L], axis) && L < R) L++;
                if (L < R) arr[R--] = arr[L];
            }
            arr[L] = piv;
            beg[i + 1] = L + 1;
            end[i + 1] = end[i];
            end[i++] = L;
        } else {
            i--;
        }
    }
    return true;
}

/* primitive.cpp:537-582: median split, random axis, sort skipped when axis repeats */
static int CreateBVHRec(BVH &bvh, PrimitiveHandle *handles, int n, int depth, int max_depth, GlibcRand &rng,
                        int last_axis) {
    int me = (int)bvh.nodes.size();
    bvh.nodes.push_back(Node());
    bvh.nodes[me].left = bvh.nodes[me].right = -1;
    bvh.nodes[me].handles = -1;
    bvh.nodes[me].n = n;
    bvh.nodes[me].is_leaf = 0;
    int axis = int(3 * rng.rand_float());
    if (axis != last_axis) {
        last_axis = axis;
        if (axis <= 2) QuickSortHandles(handles, n, axis); /* rand_float() can round to 1.0f: axis 3 sorts nothing */
    }
    if (n == 1 || depth >= max_depth) {
        bvh.nodes[me].handles = (int)bvh.handles.size();
        for (int i = 0; i < n; i++) bvh.handles.push_back(handles[i]);
        Bounds3 b = handles[0].bound;
        for (int i = 1; i < n; i++) b = Union(b, handles[i].bound);
        bvh.nodes[me].bound = b;
        bvh.nodes[me].is_leaf = 1;
        if (bvh.maxLeaf < n) bvh.maxLeaf = n;
        return me;
    }
    int l = CreateBVHRec(bvh, handles, n / 2, depth + 1, max_depth, rng, last_axis);
    int r = CreateBVHRec(bvh, &handles[n / 2], n - n / 2, depth + 1, max_depth, rng, last_axis);
    bvh.nodes[me].left = l;
    bvh.nodes[me].right = r;
    bvh.nodes[me].bound = Union(bvh.nodes[l].bound, bvh.nodes[r].bound);
    return me;
}
void BuildBVH(BVH &out, std::vector<PrimitiveHandle> &handles, int max_depth, GlibcRand &rng) {
    out.nodes.clear();
    out.handles.clear();
    out.maxLeaf = 0;
    out.nodes.reserve(2 * handles.size());
    out.handles.reserve(handles.size());
    CreateBVHRec(out, handles.data(), (int)handles.size(), 0, max_depth, rng, -1);
}

/* ---- interaction.cpp:5-39 ---------------------------------------------------------------- */
Ray SpawnRay(const Interaction &it, const V3 &d) {
    V3 o = OffsetRayOrigin(it.p, it.pError, it.n, d);
    return Ray(o, d, ORC_Infinity, 0.f, it.GetMedium(d));
}
Ray SpawnRayTo(const Interaction &it, const Interaction &to) {
    V3 origin = OffsetRayOrigin(it.p, it.pError, it.n, to.p - it.p);
    V3 target = OffsetRayOrigin(to.p, to.pError, to.n, origin - to.p);
    V3 d = target - origin;
    return Ray(origin, d, 1 - ORC_ShadowEpsilon, 0.f, it.GetMedium(d));
}
static SurfaceInteraction MakeSI(const Shape *sh, int shapeIdx, const V3 &p, const V3 &pError, const V2 &uv, const V3 &wo,
                                 const V3 &dpdu, const V3 &dpdv, const V3 &dndu, const V3 &dndv, int fid) {
    SurfaceInteraction si;
    si.p = p;
    si.n = Normalize(Cross(dpdu, dpdv));
    si.pError = pError;
    si.wo = Normalize(wo);
    si.uv = uv;
    si.dpdu = dpdu;
    si.dpdv = dpdv;
    si.dndu = dndu;
    si.dndv = dndv;
    si.shape = shapeIdx;
    si.faceIndex = fid;
    if (sh && (sh->reverseOrientation ^ sh->transformSwapsHandedness)) si.n *= -1;
    return si;
}
/* transform.cpp:224-256: note pError is NOT carried over (stays zero) */
static SurfaceInteraction XfSI(const Xf &t, const SurfaceInteraction &si) {
    SurfaceInteraction ret;
    ret.p = xf_point(t, si.p);
    ret.n = Normalize(xf_normal(t, si.n));
    ret.wo = Normalize(xf_vec(t, si.wo));
    ret.mediumInterface = si.mediumInterface;
    ret.uv = si.uv;
    ret.shape = si.shape;
    ret.dpdu = xf_vec(t, si.dpdu);
    ret.dpdv = xf_vec(t, si.dpdv);
    ret.dndu = xf_normal(t, si.dndu);
    ret.dndv = xf_normal(t, si.dndv);
    ret.primitive = si.primitive;
    ret.faceIndex = si.faceIndex;
    return ret;
}

/* ---- sphere.cpp:105-193 (also particlecloud.cpp:104-178 with a translation only) ---------- */
static bool SphereIntersectCore(const Shape *sh, int shapeIdx, const Xf &toObject, const Xf &toWorld, Float radius,
                                Float zMin, Float zMax, Float thetaMin, Float thetaMax, Float phiMax, bool cloud, int fid,
                                const Ray &r, Float *tHit, SurfaceInteraction *isect) {
    Float phi;
    V3 pHit;
    V3 oErr, dErr;
    Ray ray = xf_ray_err(toObject, r, &oErr, &dErr);
    Float ox = ray.o.x, oy = ray.o.y, oz = ray.o.z;
    Float dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
    Float a = dx * dx + dy * dy + dz * dz;
    Float b = 2 * (dx * ox + dy * oy + dz * oz);
    Float c = ox * ox + oy * oy + oz * oz - Float(radius) * Float(radius);
    Float t0, t1;
    if (!Quadratic(a, b, c, &t0, &t1)) return false;
    if (t0 > ray.tMax || t1 <= 0) return false;
    Float tShapeHit = t0;
    if (tShapeHit <= 0 || IsUnsafeHit(tShapeHit)) {
        tShapeHit = t1;
        if (tShapeHit > ray.tMax) return false;
    }
    pHit = ray.at((Float)tShapeHit);
    pHit *= radius / Distance(pHit, V3(0, 0, 0));
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;
    phi = atan2f(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * ORC_Pi;
    if (!cloud) {
        if ((zMin > -radius && pHit.z < zMin) || (zMax < radius && pHit.z > zMax) || phi > phiMax) {
            if (IsZero(tShapeHit - t1)) return false;
            if (t1 > ray.tMax) return false;
            tShapeHit = t1;
            pHit = ray.at((Float)tShapeHit);
            pHit *= radius / Distance(pHit, V3(0, 0, 0));
            if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;
            phi = atan2f(pHit.y, pHit.x);
            if (phi < 0) phi += 2 * ORC_Pi;
            if ((zMin > -radius && pHit.z < zMin) || (zMax < radius && pHit.z > zMax) || phi > phiMax) return false;
        }
    }
    Float u, v, theta;
    V3 dpdu, dpdv, d2Pduu, d2Pduv, d2Pdvv;
    theta = acosf(Clampf(pHit.z / radius, -1, 1));
    Float zRadius = sqrtf(pHit.x * pHit.x + pHit.y * pHit.y);
    Float invZRadius = 1 / zRadius;
    Float cosPhi = pHit.x * invZRadius;
    Float sinPhi = pHit.y * invZRadius;
    if (!cloud) {
        u = phi / phiMax;
        v = (theta - thetaMin) / (thetaMax - thetaMin);
        dpdu = V3(-phiMax * pHit.y, phiMax * pHit.x, 0);
        dpdv = (thetaMax - thetaMin) * V3(pHit.z * cosPhi, pHit.z * sinPhi, -radius * sinf(theta));
        d2Pduu = -phiMax * phiMax * V3(pHit.x, pHit.y, 0);
        d2Pduv = (thetaMax - thetaMin) * pHit.z * phiMax * V3(-sinPhi, cosPhi, 0.);
        d2Pdvv = -(thetaMax - thetaMin) * (thetaMax - thetaMin) * V3(pHit.x, pHit.y, pHit.z);
    } else {
        /* particlecloud.cpp:111,140-154: phiMax = 2*Pi as Float, (-Pi) narrowed when scaling a vector */
        u = phi / (2 * ORC_Pi);
        v = 1.0 - theta / ORC_Pi;
        dpdu = V3(-phiMax * pHit.y, phiMax * pHit.x, 0);
        dpdv = (Float)(-ORC_Pi) * V3(pHit.z * cosPhi, pHit.z * sinPhi, -radius * sinf(theta));
        d2Pduu = -phiMax * phiMax * V3(pHit.x, pHit.y, 0);
        d2Pduv = (Float)((-ORC_Pi) * pHit.z * phiMax) * V3(-sinPhi, cosPhi, 0.);
        d2Pdvv = (Float)((-ORC_Pi) * (ORC_Pi)) * V3(pHit.x, pHit.y, pHit.z);
    }
    Float E = Dot(dpdu, dpdu);
    Float F = Dot(dpdu, dpdv);
    Float G = Dot(dpdv, dpdv);
    V3 N = Normalize(Cross(dpdu, dpdv));
    Float e = Dot(N, d2Pduu);
    Float f = Dot(N, d2Pduv);
    Float g = Dot(N, d2Pdvv);
    Float invEGF2 = 1 / (E * G - F * F);
    V3 dndu = (f * F - e * G) * invEGF2 * dpdu + (e * F - f * E) * invEGF2 * dpdv;
    V3 dndv = (g * F - f * G) * invEGF2 * dpdu + (f * F - g * E) * invEGF2 * dpdv;
    V3 pError = gammaf(5) * Abs(pHit);
    *isect = XfSI(toWorld, MakeSI(sh, shapeIdx, pHit, pError, V2(u, v), -ray.d, dpdu, dpdv, dndu, dndv, fid));
    *tHit = (Float)tShapeHit;
    return true;
}

/* ---- rectangle.cpp:3-35 ---------------------------------------------------------------- */
static bool RectIntersect(const Shape &sh, int idx, const Ray &r, Float *tHit, SurfaceInteraction *isect) {
    V3 oErr, dErr;
    Ray ray = xf_ray_err(sh.WorldToObject, r, &oErr, &dErr);
    Float hx = sh.sizex / 2, hy = sh.sizey / 2;
    Float zb = ORC_ShadowEpsilon * Max(hx, hy);
    Bounds3 rec(V3(-hx, -hy, -zb), V3(hx, hy, zb));
    Float tmin, tfar;
    bool rv = BoundsIntersectP(rec, ray, &tmin, &tfar);
    if (!rv) return false;
    if (IsUnsafeHit(tmin)) return false;
    V3 pHit = ray.at(tmin);
    if (pHit.x < -hx || pHit.x > hx || pHit.y < -hy || pHit.y > hy) return false;
    Float u = Absf((pHit.x - hx) / sh.sizex);
    Float v = Absf((pHit.y - hy) / sh.sizey);
    V3 pError = gammaf(5) * Abs(pHit);
    *isect = XfSI(sh.ObjectToWorld,
                  MakeSI(&sh, idx, pHit, pError, V2(u, v), -ray.d, V3(1, 0, 0), V3(0, 1, 0), V3(0), V3(0), 0));
    *tHit = tmin;
    return rv;
}

/* ---- disk.cpp:8-44 --------------------------------------------------------------------- */
static bool DiskIntersect(const Shape &sh, int idx, const Ray &r, Float *tHit, SurfaceInteraction *isect) {
    V3 oErr, dErr;
    Ray ray = xf_ray_err(sh.WorldToObject, r, &oErr, &dErr);
    if (IsZero(ray.d.z)) return false;
    Float tShapeHit = (sh.height - ray.o.z) / ray.d.z;
    if (tShapeHit <= 0 || tShapeHit >= ray.tMax) return false;
    V3 pHit = ray.at(tShapeHit);
    Float dist2 = pHit.x * pHit.x + pHit.y * pHit.y;
    if (dist2 > sh.radius * sh.radius || dist2 < sh.innerRadius * sh.innerRadius) return false;
    Float phi = atan2f(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * ORC_Pi;
    if (phi > sh.phiMax) return false;
    Float u = phi / sh.phiMax;
    Float rHit = sqrtf(dist2);
    Float v = (sh.radius - rHit) / (sh.radius - sh.innerRadius);
    V3 dpdu(-sh.phiMax * pHit.y, sh.phiMax * pHit.x, 0);
    V3 dpdv = V3(pHit.x, pHit.y, 0.) * (sh.innerRadius - sh.radius) / rHit;
    pHit.z = sh.height;
    *isect = XfSI(sh.ObjectToWorld, MakeSI(&sh, idx, pHit, V3(0, 0, 0), V2(u, v), -ray.d, dpdu, dpdv, V3(0), V3(0), 0));
    *tHit = (Float)tShapeHit;
    return true;
}

/* ---- procedural.cpp -------------------------------------------------------------------- */
static Float SDFBox(const V3 &p, const V3 &boxLength) { /* :78-81 */
    V3 d = Abs(p) - boxLength;
    return Min(Max(d.x, Max(d.y, d.z)), 0.0) + Length(V3(Max(d.x, 0.f), Max(d.y, 0.f), Max(d.z, 0.f)));
}
static Float ProcMap(const Shape &sh, const V3 &p, int *shapeId) {
    switch (sh.type) {
    case SPHERE_PROCEDURAL: return Length(p) - sh.radius;          /* :74-76, 258-261 */
    case BOX_PROCEDURAL: return SDFBox(p, sh.length * 0.5);        /* :294-297 */
    default: return SDFMapToy(sh.toyId, sh.toyBounds, p, shapeId); /* :380-395 */
    }
}
static Bounds3 ProcLocalBounds(const Shape &sh) {
    if (sh.type == SPHERE_PROCEDURAL) return Bounds3(V3(-sh.radius, -sh.radius, -sh.radius), V3(sh.radius, sh.radius, sh.radius));
    if (sh.type == BOX_PROCEDURAL) {
        V3 half = sh.length * 0.5;
        return Bounds3(V3(-half.x, -half.y, -half.z), V3(half.x, half.y, half.z));
    }
    return sh.toyBounds;
}
/* :263-276, 299-352, 145-151 */
static bool ProcRefineHit(const Shape &sh, V3 *pHit, V2 *uv, int *shapeId, V3 *n) {
    if (sh.type == SPHERE_PROCEDURAL) {
        *shapeId = 0;
        *pHit *= sh.radius / Distance(*pHit, V3(0, 0, 0));
        Float phi = atan2f(pHit->y, pHit->x);
        if (phi < 0) phi += 2 * ORC_Pi;
        Float u = phi / (2 * ORC_Pi);
        Float theta = acosf(Clampf(pHit->z / sh.radius, -1, 1));
        Float v = (ORC_Pi - theta) / ORC_Pi;
        *uv = V2(u, v);
        *n = V3(pHit->x, pHit->y, pHit->z) / sh.radius;
        return true;
    }
    if (sh.type == BOX_PROCEDURAL) {
        V3 half = sh.length * 0.5;
        V3 chosen(ORC_Infinity, -1, -1);
        V3 quasi[6] = {
            V3(Absf(pHit->x - half.x), 1, +half.x), V3(Absf(pHit->x + half.x), 3, -half.x),
            V3(Absf(pHit->y - half.y), 4, +half.y), V3(Absf(pHit->y + half.y), 5, -half.y),
            V3(Absf(pHit->z - half.z), 2, +half.z), V3(Absf(pHit->z + half.z), 0, -half.z),
        };
        for (int i = 0; i < 6; i++)
            if (chosen.x > quasi[i].x) chosen = quasi[i];
        int target = (int)chosen.y;
        Float u = 0, v = 0;
        switch (target) {
        case 1: case 3:
            pHit->x = chosen.z;
            *n = chosen.z > 0 ? V3(1, 0, 0) : V3(-1, 0, 0);
            u = (pHit->z + half.z) / sh.length.z;
            v = 1.f - (pHit->y + half.y) / sh.length.y;
            break;
        case 4: case 5:
            pHit->y = chosen.z;
            *n = chosen.z > 0 ? V3(0, 1, 0) : V3(0, -1, 0);
            u = (pHit->x + half.x) / sh.length.x;
            v = (pHit->z + half.z) / sh.length.z;
            break;
        default: /* 0, 2 */
            pHit->z = chosen.z;
            *n = chosen.z > 0 ? V3(0, 0, 1) : V3(0, 0, -1);
            u = (pHit->x + half.x) / sh.length.x;
            v = 1.f - (pHit->y + half.y) / sh.length.z; /* sic: length.z */
            break;
        }
        (void)shapeId; /* left as written by the last Map() call */
        *uv = V2(u, v);
        return true;
    }
    *uv = V2(0, 0);
    return false;
}
/* :159-185 */
static V3 ProcDerivativeNormal(const Shape &sh, const V3 &p) {
    bool found = false;
    float scale = 0.001;
    V3 n;
    int it = 0, id;
    while (!found) {
        Float e = scale * Length(p);
        Float x = ProcMap(sh, V3(p.x + e, p.y, p.z), &id) - ProcMap(sh, V3(p.x - e, p.y, p.z), &id);
        Float y = ProcMap(sh, V3(p.x, p.y + e, p.z), &id) - ProcMap(sh, V3(p.x, p.y - e, p.z), &id);
        Float z = ProcMap(sh, V3(p.x, p.y, p.z + e), &id) - ProcMap(sh, V3(p.x, p.y, p.z - e), &id);
        n = V3(x, y, z);
        found = !IsZero(Length(n));
        if (!found) {
            if (it++ > 10) { n = V3(0, 1, 0); break; }
            scale *= 2;
        }
    }
    return Normalize(n);
}
/* :187-250 */
static bool ProcIntersect(const Shape &sh, int idx, const Ray &r, Float *tHit, SurfaceInteraction *isect) {
    V3 oErr, dErr;
    bool hit = false, fromInside = false;
    Float tMax;
    Float t = 0;
    V2 uv;
    int id = 0;
    Ray ur(r.o, Normalize(r.d));
    Bounds3 localBound = ProcLocalBounds(sh);
    Ray ray = xf_ray_err(sh.WorldToObject, ur, &oErr, &dErr); /* ray.tMax = Infinity here */
    ur.tMax = r.tMax * Length(r.d);
    tMax = ur.tMax;
    fromInside = Inside(ray.o, localBound);
    if (!fromInside) {
        Float t1, t2;
        if (!BoundsIntersectP(localBound, ray, &t1, &t2)) return false;
        t = Min(t1, t2);
    }
    for (int i = 0; i < 256 && t < tMax; i++) {
        V3 p = ray.at(t);
        Float d = ProcMap(sh, p, &id);
        if (Absf(d) < (0.001 * t)) { hit = true; break; }
        t += d;
        if (!Inside(ray.at(t), localBound)) break;
    }
    if (hit) {
        V3 dpdu, dpdv, n;
        V3 pHit = ray.at(t);
        if (!ProcRefineHit(sh, &pHit, &uv, &id, &n)) n = ProcDerivativeNormal(sh, pHit);
        CoordinateSystem(n, &dpdu, &dpdv);
        V3 pError = gammaf(5) * Abs(pHit);
        *isect = XfSI(sh.ObjectToWorld, MakeSI(&sh, idx, pHit, pError, uv, -ray.d, dpdu, dpdv, V3(0), V3(0), id));
        *tHit = t; /* sic: distance along the NORMALISED direction */
    }
    return hit;
}

/* ---- mesh.cpp:71-90,162-365 ------------------------------------------------------------- */
static void MeshGetUVs(const MeshData &m, V2 st[3], int triNum) {
    if (m.has_uv) {
        int i0 = m.indices[9 * triNum + 2], i1 = m.indices[9 * triNum + 5], i2 = m.indices[9 * triNum + 8];
        if ((i0 < 0 || i0 > m.nUvs) || (i1 < 0 || i1 > m.nUvs) || (i2 < 0 || i2 > m.nUvs)) {
            st[0] = V2(0, 0); st[1] = V2(1, 0); st[2] = V2(1, 1);
        } else {
            st[0] = V2(m.uv[i0].x, 1.0 - m.uv[i0].y);
            st[1] = V2(m.uv[i1].x, 1.0 - m.uv[i1].y);
            st[2] = V2(m.uv[i2].x, 1.0 - m.uv[i2].y);
        }
    } else {
        st[0] = V2(0, 0); st[1] = V2(1, 0); st[2] = V2(1, 1);
    }
}
static bool IntersectTriangle(const Shape &sh, int shapeIdx, const MeshData &m, const Ray &ray, SurfaceInteraction *isect,
                              int triNum, Float *tHit, Float *bary) {
    V3 p0 = m.p[m.indices[9 * triNum + 0]];
    V3 p1 = m.p[m.indices[9 * triNum + 3]];
    V3 p2 = m.p[m.indices[9 * triNum + 6]];
    V3 p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
    int kz = MaxDimension(Abs(ray.d));
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    V3 d = Permute(ray.d, kx, ky, kz);
    p0t = Permute(p0t, kx, ky, kz);
    p1t = Permute(p1t, kx, ky, kz);
    p2t = Permute(p2t, kx, ky, kz);
    Float Sx = -d.x / d.z, Sy = -d.y / d.z, Sz = 1.f / d.z;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (IsZero(e0) || IsZero(e1) || IsZero(e2)) {
        double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    Float det = e0 + e1 + e2;
    if (IsZero(det)) return false;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    Float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < ray.tMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > ray.tMax * det)) return false;
    Float invDet = 1 / det;
    Float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    Float t = tScaled * invDet;
    Float maxZt = MaxComponent(Abs(V3(p0t.z, p1t.z, p2t.z)));
    Float deltaZ = gammaf(3) * maxZt;
    Float maxXt = MaxComponent(Abs(V3(p0t.x, p1t.x, p2t.x)));
    Float maxYt = MaxComponent(Abs(V3(p0t.y, p1t.y, p2t.y)));
    Float deltaX = gammaf(5) * (maxXt + maxZt);
    Float deltaY = gammaf(5) * (maxYt + maxZt);
    Float deltaE = 2 * (gammaf(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    Float maxE = MaxComponent(Abs(V3(e0, e1, e2)));
    Float deltaT = 3 * (gammaf(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * Absf(invDet);
    if (t <= deltaT) return false;

    V3 dpdu, dpdv;
    V2 st[3];
    MeshGetUVs(m, st, triNum);
    V2 dst02 = st[0] - st[2], dst12 = st[1] - st[2];
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    Float determinant = dst02[0] * dst12[1] - dst02[1] * dst12[0];
    bool degenerateUV = Absf(determinant) < 1e-8;
    if (degenerateUV) {
        st[0] = V2(0, 0); st[1] = V2(1, 0); st[2] = V2(1, 1);
        dst02 = st[0] - st[2]; dst12 = st[1] - st[2];
        determinant = dst02[0] * dst12[1] - dst02[1] * dst12[0];
        degenerateUV = Absf(determinant) < 1e-8;
    }
    if (!degenerateUV) {
        Float invdet = 1 / determinant;
        dpdu = (dst12[1] * dp02 - dst02[1] * dp12) * invdet;
        dpdv = (-dst12[0] * dp02 + dst02[0] * dp12) * invdet;
    }
    Float xAbsSum = (Absf(b0 * p0.x) + Absf(b1 * p1.x) + Absf(b2 * p2.x));
    Float yAbsSum = (Absf(b0 * p0.y) + Absf(b1 * p1.y) + Absf(b2 * p2.y));
    Float zAbsSum = (Absf(b0 * p0.z) + Absf(b1 * p1.z) + Absf(b2 * p2.z));
    V3 pError = gammaf(7) * V3(xAbsSum, yAbsSum, zAbsSum);
    V3 pHit = b0 * p0 + b1 * p1 + b2 * p2;
    V2 stHit = b0 * st[0] + b1 * st[1] + b2 * st[2];
    *isect = MakeSI(&sh, shapeIdx, pHit, pError, stHit, -ray.d, dpdu, dpdv, V3(0), V3(0), triNum);
    isect->n = Normalize(Cross(dp02, dp12));
    if (sh.reverseOrientation ^ sh.transformSwapsHandedness) isect->n = -isect->n;
    *tHit = t;
    if (bary) { bary[0] = b0; bary[1] = b1; }
    return true;
}

/* shared stack traversal of primitive.cpp:374-431 / mesh.cpp:454-509 */
template <typename LeafFn> static bool TraverseRef(const Scene &sc, const BVH &bvh, const Ray &r, LeafFn leaf) {
    int stack[256];
    int *sp = stack;
    *sp++ = -1;
    int node = 0;
    bool hit_anything = false;
    while (node != -1) {
        const Node &nd = bvh.nodes[node];
        g_ctr.nodes_visited++;
        bool hit_bound = BoundsIntersectP(nd.bound, r);
        if (hit_bound) {
            if (nd.is_leaf) {
                hit_anything |= leaf(nd);
                node = *--sp;
            } else {
                Float tl0, tr0;
                g_ctr.nodes_visited += 2;
                bool visitL = BoundsIntersectP(bvh.nodes[nd.left].bound, r, &tl0);
                bool visitR = BoundsIntersectP(bvh.nodes[nd.right].bound, r, &tr0);
                if (visitR && visitL) {
                    int first, second;
                    if (tr0 < tl0) { first = nd.right; second = nd.left; }
                    else { first = nd.left; second = nd.right; }
                    *sp++ = second;
                    node = first;
                } else if (visitL) node = nd.left;
                else if (visitR) node = nd.right;
                else node = *--sp;
            }
        } else node = *--sp;
    }
    return hit_anything;
}

static bool MeshIntersect(const Scene &sc, const Shape &sh, int idx, const Ray &r, Float *tHit, SurfaceInteraction *isect,
                          Float *bary) {
    const MeshData &m = sc.meshes[sh.geom];
    bool hit = TraverseRef(sc, m.bvh, r, [&](const Node &nd) {
        bool any = false;
        for (int i = 0; i < nd.n; i++) {
            int nTri = m.bvh.handles[nd.handles + i].handle;
            g_ctr.prims_tested++;
            if (IntersectTriangle(sh, idx, m, r, isect, nTri, tHit, bary)) { any = true; r.tMax = *tHit; }
        }
        return any;
    });
    if (hit && !sc.meshfix) *isect = SurfaceInteraction(); /* mesh.cpp:504-506 clobber (as-is mode) */
    return hit;
}

/* particlecloud.cpp:197-274: the older traversal that eagerly tests leaf children */
static bool CloudIntersect(const Scene &sc, const Shape &sh, int idx, const Ray &r, Float *tHit, SurfaceInteraction *isect) {
    const CloudData &c = sc.clouds[sh.geom];
    const BVH &bvh = c.bvh;
    Float phiMax = 2 * ORC_Pi;
    auto leaf = [&](const Node &nd, SurfaceInteraction *out) {
        bool any = false;
        for (int i = 0; i < nd.n; i++) {
            int nPart = bvh.handles[nd.handles + i].handle;
            g_ctr.prims_tested++;
            Xf toWorld = XfTranslate(c.positions[nPart]);
            Xf toObject = XfInverse(toWorld);
            if (SphereIntersectCore(&sh, idx, toObject, toWorld, c.radius, 0, 0, 0, 0, phiMax, true, nPart, r, tHit, out)) {
                any = true;
                r.tMax = *tHit;
            }
        }
        return any;
    };
    int stack[256];
    int *sp = stack;
    *sp++ = -1;
    int node = 0;
    SurfaceInteraction tmp;
    bool hit_anything = false;
    Float t0, t1;
    g_ctr.nodes_visited++;
    bool hit_bound = BoundsIntersectP(bvh.nodes[node].bound, r, &t0, &t1);
    if (hit_bound && bvh.nodes[node].is_leaf) return leaf(bvh.nodes[node], isect);
    do {
        if (hit_bound) {
            const Node &nd = bvh.nodes[node];
            const Node &L = bvh.nodes[nd.left], &R = bvh.nodes[nd.right];
            bool hitl = false, hitr = false;
            if (L.n > 0 || R.n > 0) {
                g_ctr.nodes_visited += 2;
                hitl = BoundsIntersectP(L.bound, r, &t0, &t1);
                hitr = BoundsIntersectP(R.bound, r, &t0, &t1);
            }
            if (hitl && L.is_leaf) if (leaf(L, &tmp)) hit_anything = true;
            if (hitr && R.is_leaf) if (leaf(R, &tmp)) hit_anything = true;
            bool tL = (hitl && !L.is_leaf), tR = (hitr && !R.is_leaf);
            if (!tR && !tL) node = *--sp;
            else {
                node = tL ? nd.left : nd.right;
                if (tL && tR) *sp++ = nd.right;
            }
        } else node = *--sp;
        if (node != -1) { g_ctr.nodes_visited++; hit_bound = BoundsIntersectP(bvh.nodes[node].bound, r, &t0, &t1); }
    } while (node != -1);
    if (hit_anything) *isect = tmp;
    return hit_anything;
}

bool ShapeIntersect(const Scene &sc, int primIndex, const Ray &ray, Float *tHit, SurfaceInteraction *isect, Float *bary) {
    const Shape &sh = sc.prims[primIndex].shape;
    switch (sh.type) {
    case SPHERE:
        return SphereIntersectCore(&sh, primIndex, sh.WorldToObject, sh.ObjectToWorld, sh.radius, sh.zMin, sh.zMax,
                                   sh.thetaMin, sh.thetaMax, sh.phiMax, false, 0, ray, tHit, isect);
    case RECTANGLE: return RectIntersect(sh, primIndex, ray, tHit, isect);
    case DISK: return DiskIntersect(sh, primIndex, ray, tHit, isect);
    case MESH: return MeshIntersect(sc, sh, primIndex, ray, tHit, isect, bary);
    case PARTICLECLOUD: return CloudIntersect(sc, sh, primIndex, ray, tHit, isect);
    case SPHERE_PROCEDURAL: case BOX_PROCEDURAL: case COMPONENT_PROCEDURAL: return ProcIntersect(sh, primIndex, ray, tHit, isect);
    default: return false;
    }
}

Bounds3 ShapeGetBounds(const Scene &sc, const Shape &sh) {
    switch (sh.type) {
    case SPHERE: /* sphere.cpp:16-19 */
        return xf_bounds(sh.ObjectToWorld, Bounds3(V3(-sh.radius, -sh.radius, sh.zMin), V3(sh.radius, sh.radius, sh.zMax)));
    case RECTANGLE: { /* rectangle.cpp:37-40 */
        Float hx = sh.sizex / 2, hy = sh.sizey / 2;
        return xf_bounds(sh.ObjectToWorld, Bounds3(V3(-hx, -hy, -0.0001), V3(hx, hy, 0.0001)));
    }
    case DISK: /* disk.cpp:3-6 */
        return xf_bounds(sh.ObjectToWorld, Bounds3(V3(-sh.radius, -sh.radius, sh.height), V3(sh.radius, sh.radius, sh.height)));
    case MESH: return sc.meshes[sh.geom].bvh.nodes[0].bound;
    case PARTICLECLOUD: return sc.clouds[sh.geom].bvh.nodes[0].bound;
    default: return xf_bounds(sh.ObjectToWorld, ProcLocalBounds(sh));
    }
}

/* ---- areas, sampling, pdfs ---------------------------------------------------------------- */
static Float TriangleArea(const MeshData &m, int triNum) { /* mesh.cpp:377-385 */
    V3 p0 = m.p[m.indices[9 * triNum + 0]], p1 = m.p[m.indices[9 * triNum + 3]], p2 = m.p[m.indices[9 * triNum + 6]];
    return 0.5 * Length(Cross(p1 - p0, p2 - p0));
}
Float ShapeArea(const Scene &sc, const Shape &sh) {
    switch (sh.type) {
    case SPHERE: return sh.phiMax * sh.radius * (sh.zMax - sh.zMin);                                   /* sphere.cpp:21-23 */
    case RECTANGLE: return sh.sizex * sh.sizey;                                                        /* rectangle.cpp:42-44 */
    case DISK: return sh.phiMax * 0.5 * (sh.radius * sh.radius - sh.innerRadius * sh.innerRadius);      /* disk.cpp:46-48 */
    case MESH: { /* mesh.cpp:367-375 */
        const MeshData &m = sc.meshes[sh.geom];
        Float area = 0;
        for (int i = 0; i < m.nTriangles; i++) area += TriangleArea(m, i);
        return area;
    }
    case PARTICLECLOUD: { /* particlecloud.cpp:16-29 */
        const CloudData &c = sc.clouds[sh.geom];
        Float area = 0;
        for (int i = 0; i < c.nParticles; i++) area += (Float)(4 * ORC_Pi * c.radius * c.radius);
        return area;
    }
    default: return 1; /* procedural.h:59 */
    }
}

/* sphere.cpp:37-48 / particlecloud.cpp:31-45 */
static Interaction SphereSampleArea(const Scene &sc, const Shape &sh, const Xf &toWorld, Float radius, const V2 &u, Float *pdf) {
    V3 pObj = V3(0) + radius * SampleSphere(u);
    Interaction it;
    it.n = Normalize(xf_normal(toWorld, V3(pObj.x, pObj.y, pObj.z)));
    if (sh.reverseOrientation) it.n *= -1;
    pObj *= radius / Distance(pObj, V3(0, 0, 0));
    V3 pObjError = gammaf(5) * Abs(pObj);
    it.p = xf_point_err2(toWorld, pObj, pObjError, &it.pError);
    *pdf = 1 / ShapeArea(sc, sh);
    return it;
}
/* sphere.cpp:50-103 / particlecloud.cpp:47-102 */
static Interaction SphereSampleRef(const Scene &sc, const Shape &sh, const Xf &toWorld, Float radius, const Interaction &ref,
                                   const V2 &u, Float *pdf) {
    V3 pCenter = xf_point(toWorld, V3(0, 0, 0));
    V3 pOrigin = OffsetRayOrigin(ref.p, ref.pError, ref.n, pCenter - ref.p);
    if (DistanceSquared(pOrigin, pCenter) <= radius * radius) {
        Interaction intr = SphereSampleArea(sc, sh, toWorld, radius, u, pdf);
        V3 wi = intr.p - ref.p;
        if (IsZero(LengthSquared(wi))) *pdf = 0;
        else {
            wi = Normalize(wi);
            *pdf *= DistanceSquared(ref.p, intr.p) / AbsDot(intr.n, -wi);
        }
        if (std::isinf(*pdf)) *pdf = 0.f;
        return intr;
    }
    Float dc = Distance(ref.p, pCenter);
    Float invDc = 1 / dc;
    V3 wc = (pCenter - ref.p) * invDc;
    V3 wcX, wcY;
    CoordinateSystem(wc, &wcX, &wcY);
    Float sinThetaMax = radius * invDc;
    Float sinThetaMax2 = sinThetaMax * sinThetaMax;
    Float invSinThetaMax = 1 / sinThetaMax;
    Float cosThetaMax = sqrtf(Max((Float)0.f, 1 - sinThetaMax2));
    Float cosTheta = (cosThetaMax - 1) * u[0] + 1;
    Float sinTheta2 = 1 - cosTheta * cosTheta;
    if (sinThetaMax2 < 0.00068523f) {
        sinTheta2 = sinThetaMax2 * u[0];
        cosTheta = sqrtf(1 - sinTheta2);
    }
    Float cosAlpha = sinTheta2 * invSinThetaMax +
                     cosTheta * sqrtf(Max((Float)0.f, 1.f - sinTheta2 * invSinThetaMax * invSinThetaMax));
    Float sinAlpha = sqrtf(Max((Float)0.f, 1.f - cosAlpha * cosAlpha));
    Float phi = u[1] * 2 * ORC_Pi;
    V3 nWorld = SphericalDirection(sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc);
    V3 pWorld = pCenter + radius * V3(nWorld.x, nWorld.y, nWorld.z);
    Interaction it;
    it.p = pWorld;
    it.pError = gammaf(5) * Abs(pWorld);
    it.n = nWorld;
    if (sh.reverseOrientation) it.n *= -1;
    *pdf = 1 / (2 * ORC_Pi * (1 - cosThetaMax));
    return it;
}

Interaction ShapeSampleRef(const Scene &sc, const Shape &sh, const Interaction &ref, const V2 &u, Float *pdf) {
    switch (sh.type) {
    case SPHERE: return SphereSampleRef(sc, sh, sh.ObjectToWorld, sh.radius, ref, u, pdf);
    case PARTICLECLOUD: { /* uses the RECEIVER's faceIndex, particlecloud.cpp:50 */
        const CloudData &c = sc.clouds[sh.geom];
        int partNum = ref.faceIndex;
        if (partNum < 0 || partNum >= c.nParticles) partNum = 0; /* reference: out-of-bounds read */
        return SphereSampleRef(sc, sh, XfTranslate(c.positions[partNum]), c.radius, ref, u, pdf);
    }
    case RECTANGLE: case DISK: {
        Interaction intr;
        if (sh.type == RECTANGLE) { /* rectangle.cpp:50-60 */
            Float hx = sh.sizex / 2, hy = sh.sizey / 2;
            V3 pObj(-hx + u[0] * sh.sizex, -hy + u[1] * sh.sizey, 0);
            intr.n = Normalize(xf_normal(sh.ObjectToWorld, V3(0, 0, 1)));
            if (sh.reverseOrientation) intr.n *= -1;
            intr.p = xf_point_err2(sh.ObjectToWorld, pObj, V3(0, 0, 0), &intr.pError);
        } else { /* disk.cpp:50-59 */
            V2 pd = ConcentricSampleDisk(u);
            V3 pObj(pd.x * sh.radius, pd.y * sh.radius, sh.height);
            intr.n = Normalize(xf_normal(sh.ObjectToWorld, V3(0, 0, 1)));
            if (sh.reverseOrientation) intr.n *= -1;
            intr.p = xf_point_err2(sh.ObjectToWorld, pObj, V3(0, 0, 0), &intr.pError);
        }
        *pdf = 1 / ShapeArea(sc, sh);
        /* rectangle.cpp:62-77 / disk.cpp:61-76 */
        V3 wi = intr.p - ref.p;
        wi = Normalize(wi);
        if (IsZero(LengthSquared(wi))) *pdf = 0;
        else {
            wi = Normalize(wi);
            *pdf *= DistanceSquared(ref.p, intr.p) / AbsDot(intr.n, -wi);
        }
        if (std::isinf(*pdf)) *pdf = 0.f;
        return intr;
    }
    case MESH: { /* mesh.cpp:387-422, uses the RECEIVER's faceIndex */
        const MeshData &m = sc.meshes[sh.geom];
        int triNum = ref.faceIndex;
        if (triNum < 0 || triNum >= m.nTriangles) triNum = 0; /* reference: out-of-bounds read */
        V3 p0 = m.p[m.indices[9 * triNum + 0]], p1 = m.p[m.indices[9 * triNum + 3]], p2 = m.p[m.indices[9 * triNum + 6]];
        V2 b = UniformSampleTriangle(u);
        Interaction it;
        it.p = b[0] * p0 + b[1] * p1 + (1 - b[0] - b[1]) * p2;
        it.faceIndex = triNum;
        it.n = Normalize(Cross(p1 - p0, p2 - p0));
        if (sh.reverseOrientation ^ sh.transformSwapsHandedness) it.n *= -1;
        V3 pAbsSum = Abs(b[0] * p0) + Abs(b[1] * p1) + Abs((1 - b[0] - b[1]) * p2);
        it.pError = gammaf(6) * V3(pAbsSum.x, pAbsSum.y, pAbsSum.z);
        *pdf = 1.f / TriangleArea(m, triNum);
        V3 wi = it.p - ref.p;
        if (IsZero(LengthSquared(wi))) *pdf = 0;
        else {
            wi = Normalize(wi);
            *pdf *= DistanceSquared(ref.p, it.p) / AbsDot(it.n, -wi);
        }
        if (std::isinf(*pdf)) *pdf = 0;
        return it;
    }
    default: return Interaction(); /* procedural.h:65-70: pdf left untouched */
    }
}

/* sphere.cpp:3-14 Shape::Pdf */
static Float ShapePdfGeneric(const Scene &sc, int primIndex, const Interaction &ref, const V3 &wi) {
    Ray ray = SpawnRay(ref, wi);
    Float tHit;
    SurfaceInteraction isectLight;
    if (!ShapeIntersect(sc, primIndex, ray, &tHit, &isectLight, nullptr)) return 0;
    Float f = (AbsDot(isectLight.n, -wi) * ShapeArea(sc, sc.prims[primIndex].shape));
    Float pdf = DistanceSquared(ref.p, isectLight.p) / f;
    if (std::isinf(pdf)) pdf = 0.f;
    return pdf;
}
Float ShapePdfRef(const Scene &sc, int primIndex, const Interaction &ref, const V3 &wi) {
    const Shape &sh = sc.prims[primIndex].shape;
    switch (sh.type) {
    case SPHERE: { /* sphere.cpp:25-35 */
        V3 pCenter = xf_point(sh.ObjectToWorld, V3(0, 0, 0));
        V3 pOrigin = OffsetRayOrigin(ref.p, ref.pError, ref.n, pCenter - ref.p);
        if (DistanceSquared(pOrigin, pCenter) <= sh.radius * sh.radius) return ShapePdfGeneric(sc, primIndex, ref, wi);
        Float sinThetaMax2 = sh.radius * sh.radius / DistanceSquared(ref.p, pCenter);
        Float cosThetaMax = sqrtf(Max((Float)0, 1 - sinThetaMax2));
        return UniformConePdf(cosThetaMax);
    }
    case MESH: { /* mesh.cpp:424-435 */
        const MeshData &m = sc.meshes[sh.geom];
        int tri = ref.faceIndex;
        if (tri < 0 || tri >= m.nTriangles) tri = 0;
        Ray ray = SpawnRay(ref, wi);
        Float tHit;
        SurfaceInteraction isectLight;
        if (!IntersectTriangle(sh, primIndex, m, ray, &isectLight, tri, &tHit, nullptr)) return 0;
        Float f = (AbsDot(isectLight.n, -wi) * TriangleArea(m, tri));
        Float pdf = DistanceSquared(ref.p, isectLight.p) / f;
        if (std::isinf(pdf)) pdf = 0.f;
        return pdf;
    }
    case PARTICLECLOUD: return 0; /* shape.h:257-260 */
    default: return ShapePdfGeneric(sc, primIndex, ref, wi);
    }
}

/* ---- primitive.cpp:52-73, 360-461 --------------------------------------------------------- */
static bool PrimitiveIntersect(const Scene &sc, int primIndex, const Ray &ray, SurfaceInteraction *isect, Float *bary) {
    Float tHit;
    SurfaceInteraction tmp;
    Float b[2] = {0, 0};
    if (!ShapeIntersect(sc, primIndex, ray, &tHit, &tmp, b)) return false;
    *isect = tmp;
    ray.tMax = tHit;
    const Primitive &pr = sc.prims[primIndex];
    if (pr.mediumInterface.IsMediumTransition()) isect->mediumInterface = pr.mediumInterface;
    else isect->mediumInterface = MediumInterface(ray.medium);
    isect->primitive = primIndex;
    if (bary) { bary[0] = b[0]; bary[1] = b[1]; }
    return true;
}
bool SceneIntersect(const Scene &sc, const Ray &r, SurfaceInteraction *isect, Float *bary) {
    return TraverseRef(sc, sc.top, r, [&](const Node &nd) {
        bool any = false;
        for (int i = 0; i < nd.n; i++) any |= PrimitiveIntersect(sc, sc.top.handles[nd.handles + i].handle, r, isect, bary);
        return any;
    });
}
bool SceneIntersectTr(const Scene &sc, Ray ray, SurfaceInteraction *isect, V3 *Tr) {
    *Tr = V3(1);
    while (true) {
        g_ctr.mis_rays++;
        bool hitSurface = SceneIntersect(sc, ray, isect);
        if (ray.medium >= 0) *Tr *= MediumTr(sc.media[ray.medium], ray);
        if (!hitSurface || sc.prims[isect->primitive].emissive) return false;
        if (sc.prims[isect->primitive].material >= 0) return true;
        ray = SpawnRay(*isect, ray.d);
    }
}

/* ---- procedural.cpp:100-130,398-574 toys: see orc_toys.cpp ------------------------------- */

} // namespace orc
