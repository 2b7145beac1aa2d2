/*
 * dev_light.h — lights, light-map distributions, shape sampling/pdf, homogeneous media.
 * Reference: src/core/light.cpp, src/lights/diffusearea.cpp, src/lights/infinite.cpp:41-91,
 * src/core/texture.h:132-179 (MipMap level 0), src/core/sampling.h:19-77, src/core/medium.cpp,
 * Shape::Sample / Shape::Pdf of src/shapes/{sphere,rectangle,disk,mesh,particlecloud}.cpp.
 */
#pragma once
#include "dev_bsdf.h"

/* the Interaction fields direct lighting reads; surface and medium interactions share it
 * (a medium interaction has n = pError = 0, interaction.h:77-99) */
struct Ref {
    f3 p, pError, n, wo;
    int face, med_in, med_out;
};
PD Ref ref_from_si(const SI &si) {
    Ref r;
    r.p = si.p; r.pError = si.pError; r.n = si.n; r.wo = si.wo; r.face = si.face; r.med_in = si.med_in; r.med_out = si.med_out;
    return r;
}
PD int ref_medium(const Ref &r, const f3 &w) { return si_medium(r.n, r.med_in, r.med_out, w); }

/* ---- texture.h:132-179, wrap = Repeat, level 0 ---- */
PD int mod_i(int a, int b) { int result = a - (a / b) * b; return (result < 0) ? result + b : result; }
PD f3 env_texel(const DScene &sc, int s, int t) {
    s = mod_i(s, sc.env_w);
    t = mod_i(t, sc.env_h);
    const float *p = sc.env_texels + 3 * (size_t)(s + t * sc.env_w);
    return f3(p[0], p[1], p[2]);
}
PD f3 env_lookup(const DScene &sc, const f2 &st) {
    float s = st.x * sc.env_w - 0.5f;
    float t = st.y * sc.env_h - 0.5f;
    int s0 = (int)floorf(s), t0 = (int)floorf(t);
    float ds = s - s0, dt = t - t0;
    return (1 - ds) * (1 - dt) * env_texel(sc, s0, t0) + (1 - ds) * dt * env_texel(sc, s0, t0 + 1) +
           ds * (1 - dt) * env_texel(sc, s0 + 1, t0) + ds * dt * env_texel(sc, s0 + 1, t0 + 1);
}

/* ---- sampling.h:19-77 ---- */
/* geometry.h:1091-1107 with pred cdf[i] <= u: the number of leading entries <= u, minus one, clamped.
 * The reference bisects (10-11 dependent loads per lookup on a 512 x 1024 light map, two lookups
 * per light sample: the longest dependent-load chain of the shade kernel).  The entries are
 * non-decreasing, so the partition point is unique and an 8-ary search finds the same index with
 * seven independent loads per round: 4 rounds instead of 11. */
PD int find_interval_cdf(const float *cdf, int size, float u) {
    int lo = 0, hi = size; /* cdf[i] <= u for all i < lo, cdf[i] > u for all i >= hi */
    while (hi - lo > 7) {
        const int step = (hi - lo) >> 3;
        int c = 0;
#pragma unroll
        for (int j = 1; j <= 7; j++) c += cdf[lo + j * step] <= u ? 1 : 0;
        const int base = lo;
        if (c < 7) hi = base + (c + 1) * step;
        if (c > 0) lo = base + c * step + 1;
    }
    int c = 0;
#pragma unroll
    for (int j = 0; j < 7; j++) c += (lo + j < hi && cdf[lo + j] <= u) ? 1 : 0;
    return clampi(lo + c - 1, 0, size - 2);
}
/* The same index from a 16-ary search tree over cdf[0 .. size) (round 3; built at upload, ptr_runtime.hip build_cdf_tree): a node
 * is one 64-byte line of 15 separators, level L has 16^L nodes, the separators of a node that covers [s, s + 16 B) are the
 * entries s + B, s + 2 B, ..., s + 15 B (+inf beyond the array).  A round reads ONE line where the search above reads seven, and
 * 1024 entries take three dependent rounds instead of four: light sampling is the longest chain of dependent loads of the shade
 * kernel.  Entry s of every node visited is <= u (it was a separator that was, or it is cdf[0] = 0), so the
 * count of entries <= u is s + c + 1 at the last level: the same number the bisection finds. */
#define CDF_TREE_FAN 16
PD int find_interval_tree(const float *tree, int levels, int size, float u) {
    int node = 0, s = 0, first = 0, width = 1; /* node within its level, start of its range, first node of the level, nodes in the level */
    int B = 1;
    for (int l = 1; l < levels; l++) B *= CDF_TREE_FAN;
    int c = 0;
    for (int l = 0; l < levels; l++) {
        const float4 *np = reinterpret_cast<const float4 *>(tree + (size_t)(first + node) * CDF_TREE_FAN);
        const float4 k0 = np[0], k1 = np[1], k2 = np[2], k3 = np[3];
        c = (k0.x <= u) + (k0.y <= u) + (k0.z <= u) + (k0.w <= u) + (k1.x <= u) + (k1.y <= u) + (k1.z <= u) + (k1.w <= u) + (k2.x <= u) +
            (k2.y <= u) + (k2.z <= u) + (k2.w <= u) + (k3.x <= u) + (k3.y <= u) + (k3.z <= u);
        s += c * B;
        first += width;
        width *= CDF_TREE_FAN;
        node = node * CDF_TREE_FAN + c;
        B /= CDF_TREE_FAN;
    }
    return clampi(s, 0, size - 2); /* (count - 1 = s + c' + 1 - 1 with the last level's c already added to s) */
}
PD float d1_sample(const float *func, const float *cdf, float funcInt, int size, float u, float *pdf, int *off, const float *tree = nullptr,
                   int tree_levels = 0) {
    int offset = tree ? find_interval_tree(tree, tree_levels, size, u) : find_interval_cdf(cdf, size, u);
    if (off) *off = offset;
    float du = u - cdf[offset];
    if ((cdf[offset + 1] - cdf[offset]) > 0) du /= (cdf[offset + 1] - cdf[offset]);
    if (pdf) *pdf = (funcInt > 0) ? func[offset] / funcInt : 0;
    return (offset + du) / size;
}
PD f2 d2_sample(const DScene &sc, const f2 &u, float *pdf) {
    float p0, p1;
    int v;
    float d1 = d1_sample(sc.env_func_int, sc.env_marg_cdf, sc.env_marg_int, sc.env_nv, u.y, &p1, &v, sc.env_marg_tree, sc.env_tree_v_levels);
    float d0 = d1_sample(sc.env_func + (size_t)v * sc.env_nu, sc.env_cdf + (size_t)v * (sc.env_nu + 1), sc.env_func_int[v], sc.env_nu, u.x, &p0,
                         nullptr, sc.env_cdf_tree ? sc.env_cdf_tree + (size_t)v * sc.env_tree_u_nodes * CDF_TREE_FAN : nullptr, sc.env_tree_u_levels);
    *pdf = p0 * p1;
    return f2(d0, d1);
}
PD float d2_pdf(const DScene &sc, const f2 &p) {
    int iu = clampi(int(p.x * sc.env_nu), 0, sc.env_nu - 1);
    int iv = clampi(int(p.y * sc.env_nv), 0, sc.env_nv - 1);
    return sc.env_func[(size_t)iv * sc.env_nu + iu] / sc.env_marg_int;
}

/* ---- light radiance ---- */
PD f3 light_le(const DScene &sc, const DLight &l, const f3 &rd) { /* infinite.cpp:58-62 */
    if (l.type != LT_INFINITE) return f3(0.f);
    f3 w = normalize(xf_vec(l.w2l, rd));
    f2 st(spherical_phi(w) * kInv2Pi, spherical_theta(w) * kInvPi);
    return env_lookup(sc, st);
}
PD f3 sum_lights_le(const DScene &sc, const f3 &rd, const f3 &beta, f3 L) {
    for (int i = 0; i < sc.n_lights; i++) L += beta * light_le(sc, sc.lights[i], rd);
    return L;
}
/* SurfaceInteraction::Le -> DiffuseAreaLight::L with twoSided = true (interaction.cpp:50-60,
 * diffusearea.cpp:5-7, scene.cpp:953) */
PD f3 surface_le(const DScene &sc, int prim) {
    int li = sc.prims[prim].light;
    if (li < 0) return f3(0.f);
    const DLight &l = sc.lights[li];
    if (l.type != LT_AREA) return f3(0.f);
    return f3(l.Lemit[0], l.Lemit[1], l.Lemit[2]);
}

/* ---- Shape::Sample(ref, u, pdf) ---- */
struct LightPoint { f3 p, pError, n; };

PD void sphere_sample_area(const DPrim &pr, const float *o2w, const float *w2o, float radius, float area, const f2 &u, LightPoint *it,
                           float *pdf) {
    f3 pObj = f3(0.f) + radius * sample_sphere(u);
    it->n = normalize(xf_normal(w2o, pObj));
    if (pr.flags & PF_REVERSE) it->n *= -1.0f;
    pObj *= radius / dist(pObj, f3(0, 0, 0));
    f3 pObjError = gammar(5) * abs3(pObj);
    it->p = xf_point_err2(o2w, pObj, pObjError, &it->pError);
    *pdf = 1 / area;
}
PD void sphere_sample_ref(const DPrim &pr, const float *o2w, const float *w2o, float radius, float area, const Ref &ref, const f2 &u,
                          LightPoint *it, float *pdf) {
    f3 pCenter = xf_point(o2w, f3(0, 0, 0));
    f3 pOrigin = offset_ray_origin(ref.p, ref.pError, ref.n, pCenter - ref.p);
    if (dist2(pOrigin, pCenter) <= radius * radius) {
        sphere_sample_area(pr, o2w, w2o, radius, area, u, it, pdf);
        f3 wi = it->p - ref.p;
        if (is_zero(length2(wi))) *pdf = 0;
        else {
            wi = normalize(wi);
            *pdf *= dist2(ref.p, it->p) / absdot(it->n, -wi);
        }
        if (isinf(*pdf)) *pdf = 0.f;
        return;
    }
    float dc = dist(ref.p, pCenter);
    float invDc = 1 / dc;
    f3 wc = (pCenter - ref.p) * invDc;
    f3 wcX, wcY;
    coordinate_system(wc, &wcX, &wcY);
    float sinThetaMax = radius * invDc;
    float sinThetaMax2 = sinThetaMax * sinThetaMax;
    float invSinThetaMax = 1 / sinThetaMax;
    float cosThetaMax = sqrtf(fmaxr(0.f, 1 - sinThetaMax2));
    float cosTheta = (cosThetaMax - 1) * u.x + 1;
    float sinTheta2 = 1 - cosTheta * cosTheta;
    if (sinThetaMax2 < 0.00068523f) {
        sinTheta2 = sinThetaMax2 * u.x;
        cosTheta = sqrtf(1 - sinTheta2);
    }
    float cosAlpha = sinTheta2 * invSinThetaMax + cosTheta * sqrtf(fmaxr(0.f, 1.f - sinTheta2 * invSinThetaMax * invSinThetaMax));
    float sinAlpha = sqrtf(fmaxr(0.f, 1.f - cosAlpha * cosAlpha));
    float phi = u.y * 2 * kPi;
    f3 nWorld = spherical_direction(sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc);
    f3 pWorld = pCenter + radius * nWorld;
    it->p = pWorld;
    it->pError = gammar(5) * abs3(pWorld);
    it->n = nWorld;
    if (pr.flags & PF_REVERSE) it->n *= -1.0f;
    *pdf = 1 / (2 * kPi * (1 - cosThetaMax));
}

PD void shape_sample_ref(const DScene &sc, int prim, const Ref &ref, const f2 &u, LightPoint *it, float *pdf) {
    const DPrim &pr = sc.prims[prim];
    it->p = f3(0.f); it->pError = f3(0.f); it->n = f3(0.f);
    switch (pr.type) {
    case SH_SPHERE: sphere_sample_ref(pr, pr.o2w, pr.w2o, pr.p[0], pr.area, ref, u, it, pdf); return;
    case SH_CLOUD: { /* uses the RECEIVER's faceIndex, particlecloud.cpp:50 */
        const DCloud &c = sc.clouds[pr.geom];
        if (c.n <= 0) { *pdf = 0; return; } /* an emitter without particles: nothing to sample (the reference would read past its array) */
        int part = ref.face;
        if (part < 0 || part >= c.n) part = 0;
        f3 ctr(c.pos[3 * part], c.pos[3 * part + 1], c.pos[3 * part + 2]);
        float o2w[12] = {1, 0, 0, ctr.x, 0, 1, 0, ctr.y, 0, 0, 1, ctr.z};
        float w2o[12] = {1, 0, 0, -ctr.x, 0, 1, 0, -ctr.y, 0, 0, 1, -ctr.z};
        sphere_sample_ref(pr, o2w, w2o, c.radius, pr.area, ref, u, it, pdf);
        return;
    }
    case SH_RECT: case SH_DISK: {
        f3 pObj;
        if (pr.type == SH_RECT) { /* rectangle.cpp:50-60 */
            float hx = pr.p[0] / 2, hy = pr.p[1] / 2;
            pObj = f3(-hx + u.x * pr.p[0], -hy + u.y * pr.p[1], 0);
        } else { /* disk.cpp:50-59 */
            f2 pd = concentric_sample_disk(u);
            pObj = f3(pd.x * pr.p[1], pd.y * pr.p[1], pr.p[0]);
        }
        it->n = normalize(xf_normal(pr.w2o, f3(0, 0, 1)));
        if (pr.flags & PF_REVERSE) it->n *= -1.0f;
        it->p = xf_point_err2(pr.o2w, pObj, f3(0, 0, 0), &it->pError);
        *pdf = 1 / pr.area;
        f3 wi = it->p - ref.p;
        wi = normalize(wi);
        if (is_zero(length2(wi))) *pdf = 0;
        else {
            wi = normalize(wi);
            *pdf *= dist2(ref.p, it->p) / absdot(it->n, -wi);
        }
        if (isinf(*pdf)) *pdf = 0.f;
        return;
    }
    case SH_MESH: { /* mesh.cpp:387-422, uses the RECEIVER's faceIndex */
        const DMesh &m = sc.meshes[pr.geom];
        if (m.n_tris <= 0) { *pdf = 0; return; } /* an emitter without triangles */
        int tri = ref.face;
        if (tri < 0 || tri >= m.n_tris) tri = 0;
        f3 p0, p1, p2;
        tri_vertices(sc, pr, tri, &p0, &p1, &p2);
        f2 b = uniform_sample_triangle(u);
        it->p = b.x * p0 + b.y * p1 + (1 - b.x - b.y) * p2;
        it->n = normalize(cross(p1 - p0, p2 - p0));
        if (pr.flags & PF_FLIP_N) it->n *= -1.0f;
        f3 pAbsSum = abs3(b.x * p0) + abs3(b.y * p1) + abs3((1 - b.x - b.y) * p2);
        it->pError = gammar(6) * pAbsSum;
        float area = 0.5 * length(cross(p1 - p0, p2 - p0));
        *pdf = 1.f / area;
        f3 wi = it->p - ref.p;
        if (is_zero(length2(wi))) *pdf = 0;
        else {
            wi = normalize(wi);
            *pdf *= dist2(ref.p, it->p) / absdot(it->n, -wi);
        }
        if (isinf(*pdf)) *pdf = 0;
        return;
    }
    default: return; /* procedural.h:65-70: pdf left untouched */
    }
}

PD f3 spawn_origin(const Ref &r, const f3 &d) { return offset_ray_origin(r.p, r.pError, r.n, d); }

PD float shape_pdf_generic(const DScene &sc, int prim, const Ref &ref, const f3 &wi) {
    f3 ro = spawn_origin(ref, wi);
    float t;
    int face;
    if (!shape_hit_t(sc, sc.prims[prim], ro, wi, kInfinity, &t, &face)) return 0; /* Shape::Pdf, sphere.cpp:3-14 */
    SI li;
    make_si(sc, prim, face, ro, wi, t, -1, &li);
    float f = (absdot(li.n, -wi) * sc.prims[prim].area);
    float pdf = dist2(ref.p, li.p) / f;
    if (isinf(pdf)) pdf = 0.f;
    return pdf;
}
PD float shape_pdf_ref(const DScene &sc, int prim, const Ref &ref, const f3 &wi) {
    const DPrim &pr = sc.prims[prim];
    switch (pr.type) {
    case SH_SPHERE: { /* sphere.cpp:25-35 */
        f3 pCenter = xf_point(pr.o2w, f3(0, 0, 0));
        f3 pOrigin = offset_ray_origin(ref.p, ref.pError, ref.n, pCenter - ref.p);
        if (dist2(pOrigin, pCenter) <= pr.p[0] * pr.p[0]) return shape_pdf_generic(sc, prim, ref, wi);
        float sinThetaMax2 = pr.p[0] * pr.p[0] / dist2(ref.p, pCenter);
        float cosThetaMax = sqrtf(fmaxr(0.f, 1 - sinThetaMax2));
        return uniform_cone_pdf(cosThetaMax);
    }
    case SH_MESH: { /* mesh.cpp:424-435 */
        const DMesh &m = sc.meshes[pr.geom];
        if (m.n_tris <= 0) return 0;
        int tri = ref.face;
        if (tri < 0 || tri >= m.n_tris) tri = 0;
        f3 ro = spawn_origin(ref, wi);
        SI li;
        if (!tri_si(sc, pr, ro, wi, tri, &li)) return 0;
        f3 p0, p1, p2;
        tri_vertices(sc, pr, tri, &p0, &p1, &p2);
        float area = 0.5 * length(cross(p1 - p0, p2 - p0));
        float f = (absdot(li.n, -wi) * area);
        float pdf = dist2(ref.p, li.p) / f;
        if (isinf(pdf)) pdf = 0.f;
        return pdf;
    }
    case SH_CLOUD: return 0; /* shape.h:257-260 */
    default: return shape_pdf_generic(sc, prim, ref, wi);
    }
}

/* ---- Light::Sample_Li / Pdf_Li ---- */
PD f3 light_sample_li(const DScene &sc, const DLight &l, const Ref &ref, const f2 &u, f3 *wi, float *pdf, LightPoint *p1) {
    if (l.type == LT_AREA) { /* diffusearea.cpp:15-28 */
        shape_sample_ref(sc, l.prim, ref, u, p1, pdf);
        if (is_zero(*pdf) || is_zero(length2(p1->p - ref.p))) { *pdf = 0; return f3(0.f); }
        *wi = normalize(p1->p - ref.p);
        return f3(l.Lemit[0], l.Lemit[1], l.Lemit[2]);
    }
    if (l.type == LT_INFINITE) { /* infinite.cpp:64-83 */
        float mapPdf;
        f2 uv = d2_sample(sc, u, &mapPdf);
        if (is_zero(mapPdf)) return f3(0.f);
        float theta = uv.y * kPi, phi = uv.x * 2 * kPi;
        float cosTheta = cosf(theta), sinTheta = sinf(theta);
        float sinPhi = sinf(phi), cosPhi = cosf(phi);
        *wi = xf_vec(l.l2w, f3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
        *pdf = mapPdf / (2 * kPi * kPi * sinTheta);
        if (is_zero(sinTheta)) *pdf = 0;
        p1->p = ref.p + *wi * (2 * l.worldRadius);
        p1->pError = f3(0.f);
        p1->n = f3(0.f);
        return env_lookup(sc, uv);
    }
    if (l.type == LT_DISTANT) { /* distant.cpp:19-28 */
        const f3 wl(l.wLight[0], l.wLight[1], l.wLight[2]);
        *wi = wl;
        *pdf = 1;
        p1->p = ref.p + wl * 2 * l.worldRadius;
        p1->pError = f3(0.f);
        p1->n = f3(0.f);
        return f3(l.Lemit[0], l.Lemit[1], l.Lemit[2]);
    }
    *pdf = 0;
    return f3(0.f);
}
PD bool light_is_delta(const DLight &l) { return l.type == LT_DISTANT; } /* IsDeltaLight(flags), light.h:39-41: DeltaDirection */
PD float light_pdf_li(const DScene &sc, const DLight &l, const Ref &ref, const f3 &w) {
    if (l.type == LT_AREA) return shape_pdf_ref(sc, l.prim, ref, w);
    if (l.type == LT_INFINITE) { /* infinite.cpp:85-91 */
        f3 wi = xf_vec(l.w2l, w);
        float theta = spherical_theta(wi), phi = spherical_phi(wi);
        float sinTheta = sinf(theta);
        if (is_zero(sinTheta)) return 0;
        return d2_pdf(sc, f2(phi * kInv2Pi, theta * kInvPi)) / (2 * kPi * kPi * sinTheta);
    }
    return 0; /* distant.cpp:30-32 */
}

/* ---- medium.cpp ---- */
PD float phase_hg(float cosTheta, float g) {
    float denom = 1 + g * g + 2 * g * cosTheta;
    return kInv4Pi * (1 - g * g) / (denom * sqrtf(denom));
}
PD float phase_sample_p(float g, const f3 &wo, f3 *wi, const f2 &u) {
    float cosTheta;
    if (absr(g) < 1e-3) cosTheta = 1 - 2 * u.x;
    else {
        float sqrTerm = (1 - g * g) / (1 - g + 2 * g * u.x);
        cosTheta = (1 + g * g - sqrTerm * sqrTerm) / (2 * g);
    }
    float sinTheta = sqrtf(fmaxr(0.f, 1 - cosTheta * cosTheta));
    float phi = 2 * kPi * u.y;
    f3 v1, v2;
    coordinate_system(wo, &v1, &v2);
    *wi = spherical_direction(sinTheta, cosTheta, phi, v1, v2, -wo);
    return phase_hg(-cosTheta, g);
}
PD f3 med_sigma_t(const DMedium &m) { return f3(m.sigma_t[0], m.sigma_t[1], m.sigma_t[2]); }
PD f3 medium_tr(const DMedium &m, const f3 &rd, float tmax) { /* medium.cpp:66-68 */
    return exp3(-med_sigma_t(m) * fminr(tmax * length(rd), FLT_MAX));
}
/* medium.cpp:70-100; returns the throughput factor, *sampled / *pm / *g describe the interaction */
PD f3 medium_sample(const DMedium &m, const f3 &ro, const f3 &rd, float tmax, const f2 &u, bool *sampled, f3 *pm, float *gOut) {
    int nn = 3;
    int channel = (int)fminr((float)(int)(u.x * nn), (float)(nn - 1));
    float distv = -logf(1 - u.y) / m.sigma_t[channel];
    float len = length(rd);
    float t = fminr(distv / len, tmax);
    bool sampledMedium = t < tmax;
    *sampled = sampledMedium;
    if (sampledMedium) {
        *pm = ro + rd * t;
        float g = m.g;
        if (g > 1 || g < -1) { g = g > 1 ? .999 : g; g = g < -1 ? -.999 : g; }
        *gOut = g;
    }
    f3 sig = med_sigma_t(m);
    f3 Tr = exp3(-sig * fminr(t, FLT_MAX) * length(rd));
    f3 density = sampledMedium ? (sig * Tr) : Tr;
    float pdf = 0;
    pdf += density.x; pdf += density.y; pdf += density.z;
    pdf *= 1 / (float)nn;
    if (is_zero(pdf)) pdf = 1;
    return sampledMedium ? (Tr * f3(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]) / pdf) : (Tr / pdf);
}
