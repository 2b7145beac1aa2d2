/*
 * orc_light.cpp — ORACLE (test infrastructure only).  PARITY UNPINNED.
 * Restates: lights (src/core/light.cpp:5-40, src/lights/diffusearea.cpp, distant.cpp,
 * infinite.cpp:41-91), MipMap level-0 lookup (src/core/texture.h:132-179), Distribution1D/2D
 * (src/core/sampling.h:19-77), direct lighting (src/core/primitive.cpp:246-358) and
 * homogeneous media (src/core/medium.cpp:4-100).
 */
#include "orc_scene.h"

namespace orc {

/* ---- texture.h:132-179: MipMap<Spectrum>, wrap = Repeat, only level 0 reachable --------------- */
static int ModI(int a, int b) { /* geometry.h:1083-1086 */
    int result = a - (a / b) * b;
    return (int)((result < 0) ? result + b : result);
}
static const V3 &EnvTexel(const EnvMap &env, int s, int t) {
    s = ModI(s, env.width);
    t = ModI(t, env.height);
    return env.texels[s + t * env.width];
}
V3 EnvLookup(const EnvMap &env, const V2 &st) { /* Lookup(st, 0) -> triangle(0, st) */
    Float s = st[0] * env.width - 0.5f;
    Float t = st[1] * env.height - 0.5f;
    int s0 = std::floor(s), t0 = std::floor(t);
    Float ds = s - s0, dt = t - t0;
    return (1 - ds) * (1 - dt) * EnvTexel(env, s0, t0) + (1 - ds) * dt * EnvTexel(env, s0, t0 + 1) +
           ds * (1 - dt) * EnvTexel(env, s0 + 1, t0) + ds * dt * EnvTexel(env, s0 + 1, t0 + 1);
}

/* ---- sampling.h:19-77 ------------------------------------------------------------------------------ */
static Float D1_SampleContinuous(const Float *func, const Float *cdf, Float funcInt, int size, Float u, Float *pdf, int *off) {
    int offset = FindInterval(size, [&](int index) { return cdf[index] <= u; });
    if (off) *off = offset;
    Float du = u - cdf[offset];
    if ((cdf[offset + 1] - cdf[offset]) > 0) du /= (cdf[offset + 1] - cdf[offset]);
    if (pdf) *pdf = (funcInt > 0) ? func[offset] / funcInt : 0;
    return (offset + du) / size;
}
static V2 D2_SampleContinuous(const EnvMap &e, const V2 &u, Float *pdf) {
    Float pdfs[2];
    int v;
    Float d1 = D1_SampleContinuous(e.funcInt.data(), e.margCdf.data(), e.margInt, e.nv, u[1], &pdfs[1], &v);
    Float d0 = D1_SampleContinuous(&e.func[(size_t)v * e.nu], &e.cdf[(size_t)v * (e.nu + 1)], e.funcInt[v], e.nu, u[0], &pdfs[0],
                                   nullptr);
    *pdf = pdfs[0] * pdfs[1];
    return V2(d0, d1);
}
static Float D2_Pdf(const EnvMap &e, const V2 &p) {
    int iu = Clampi(int(p[0] * e.nu), 0, e.nu - 1);
    int iv = Clampi(int(p[1] * e.nv), 0, e.nv - 1);
    return e.func[(size_t)iv * e.nu + iu] / e.margInt;
}

/* ---- light dispatch (light.cpp:65-205) --------------------------------------------------------------- */
V3 LightLe(const Scene &sc, const Light &l, const Ray &r) {
    if (l.type != Infinite) return V3(0); /* diffusearea.cpp:11-13, distant.cpp:15-17 */
    V3 w = Normalize(xf_vec(l.WorldToLight, r.d)); /* infinite.cpp:58-62 */
    V2 st(SphericalPhi(w) * ORC_Inv2Pi, SphericalTheta(w) * ORC_InvPi);
    return EnvLookup(sc.env, st);
}
V3 LightL(const Light &l, const Interaction &intr, const V3 &w) {
    if (l.type != DiffuseArea) return V3(0);
    return (l.twoSided || Dot(intr.n, w) > 0) ? l.Lemit : V3(0.f); /* diffusearea.cpp:5-7 */
}
V3 SurfaceLe(const Scene &sc, const SurfaceInteraction &si, const V3 &w) { /* interaction.cpp:50-60 */
    int li = sc.prims[si.primitive].light;
    return li >= 0 ? LightL(sc.lights[li], si, w) : V3(0.f);
}

struct VisibilityTester {
    Interaction p0, p1;
};
/* light.cpp:5-8 */
static bool Unoccluded(const Scene &sc, const VisibilityTester &v) {
    SurfaceInteraction tmp;
    g_ctr.shadow_rays++;
    return !SceneIntersect(sc, SpawnRayTo(v.p0, v.p1), &tmp);
}
/* light.cpp:10-40 */
static V3 VisTr(const Scene &sc, const VisibilityTester &v) {
    Ray ray(SpawnRayTo(v.p0, v.p1));
    V3 Tr(1.f);
    while (true) {
        SurfaceInteraction isect;
        g_ctr.shadow_rays++;
        bool hitSurface = SceneIntersect(sc, ray, &isect);
        if (hitSurface && sc.prims[isect.primitive].material >= 0) return V3(0.0f);
        if (ray.medium >= 0) Tr *= MediumTr(sc.media[ray.medium], ray);
        if (!hitSurface || sc.prims[isect.primitive].emissive) break;
        ray = SpawnRayTo(isect, v.p1);
    }
    return Tr;
}

static V3 LightSample_Li(const Scene &sc, const Light &l, const Interaction &ref, const V2 &u, V3 *wi, Float *pdf,
                         VisibilityTester *vis) {
    switch (l.type) {
    case DiffuseArea: { /* diffusearea.cpp:15-28 */
        Interaction pShape = ShapeSampleRef(sc, sc.prims[l.prim].shape, ref, u, pdf);
        if (IsZero(*pdf) || IsZero(LengthSquared(pShape.p - ref.p))) { *pdf = 0; return V3(0.f); }
        *wi = Normalize(pShape.p - ref.p);
        vis->p0 = ref;
        vis->p1 = pShape;
        return LightL(l, pShape, -*wi);
    }
    case Infinite: { /* infinite.cpp:64-83 */
        Float mapPdf;
        V2 uv = D2_SampleContinuous(sc.env, u, &mapPdf);
        if (IsZero(mapPdf)) return V3(0.f);
        Float theta = uv[1] * ORC_Pi, phi = uv[0] * 2 * ORC_Pi;
        Float cosTheta = cosf(theta), sinTheta = sinf(theta);
        Float sinPhi = sinf(phi), cosPhi = cosf(phi);
        *wi = xf_vec(l.LightToWorld, V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
        *pdf = mapPdf / (2 * ORC_Pi * ORC_Pi * sinTheta);
        if (IsZero(sinTheta)) *pdf = 0;
        vis->p0 = ref;
        Interaction far;
        far.p = ref.p + *wi * (2 * l.worldRadius);
        vis->p1 = far;
        return EnvLookup(sc.env, uv);
    }
    case Distant: { /* distant.cpp:19-28 */
        *wi = l.wLight;
        *pdf = 1;
        vis->p0 = ref;
        Interaction far;
        far.p = ref.p + l.wLight * 2 * l.worldRadius;
        vis->p1 = far;
        return l.Lemit;
    }
    default: *pdf = 0; return V3(0);
    }
}
static Float LightPdf_Li(const Scene &sc, const Light &l, const Interaction &ref, const V3 &w) {
    switch (l.type) {
    case DiffuseArea: return ShapePdfRef(sc, l.prim, ref, w); /* diffusearea.cpp:30-32 */
    case Infinite: { /* infinite.cpp:85-91 */
        V3 wi = xf_vec(l.WorldToLight, w);
        Float theta = SphericalTheta(wi), phi = SphericalPhi(wi);
        Float sinTheta = sinf(theta);
        if (IsZero(sinTheta)) return 0;
        return D2_Pdf(sc.env, V2(phi * ORC_Inv2Pi, theta * ORC_InvPi)) / (2 * ORC_Pi * ORC_Pi * sinTheta);
    }
    default: return 0;
    }
}
static bool IsDeltaLight(int flags) { return (flags & LF_DeltaPosition) || (flags & LF_DeltaDirection); }

/* ---- medium.cpp --------------------------------------------------------------------------------------- */
static Float PhaseHG(Float cosTheta, Float g) {
    Float denom = 1 + g * g + 2 * g * cosTheta;
    return ORC_Inv4Pi * (1 - g * g) / (denom * sqrtf(denom));
}
Float PhaseP(const PhaseFunction &ph, const V3 &wo, const V3 &wi) { return PhaseHG(Dot(wo, wi), ph.g); }
Float PhaseSample_p(const PhaseFunction &ph, const V3 &wo, V3 *wi, const V2 &u) {
    Float g = ph.g;
    Float cosTheta;
    if (Absf(g) < 1e-3) cosTheta = 1 - 2 * u[0];
    else {
        Float sqrTerm = (1 - g * g) / (1 - g + 2 * g * u[0]);
        cosTheta = (1 + g * g - sqrTerm * sqrTerm) / (2 * g);
    }
    Float sinTheta = sqrtf(Max((Float)0, 1 - cosTheta * cosTheta));
    Float phi = 2 * ORC_Pi * u[1];
    V3 v1, v2;
    CoordinateSystem(wo, &v1, &v2);
    *wi = SphericalDirection(sinTheta, cosTheta, phi, v1, v2, -wo);
    return PhaseHG(-cosTheta, g);
}
V3 MediumTr(const Medium &m, const Ray &ray) { /* medium.cpp:66-68 */
    return ExpV(-m.sigma_t * Min(ray.tMax * Length(ray.d), FLT_MAX));
}
V3 MediumSample(const Scene &sc, int medium, const Ray &ray, const V2 &u, MediumInteraction *mi) { /* medium.cpp:70-100 */
    const Medium &m = sc.media[medium];
    int nn = 3;
    int channel = (int)Min((Float)(int)(u[0] * nn), (Float)(nn - 1));
    Float dist = -logf(1 - u[1]) / m.sigma_t[channel];
    Float len = Length(ray.d);
    Float t = Min(dist / len, ray.tMax);
    bool sampledMedium = t < ray.tMax;
    if (sampledMedium) {
        *mi = MediumInteraction();
        mi->p = ray.at(t);
        mi->wo = -ray.d;
        mi->mediumInterface = MediumInterface(medium);
        Float g = m.g; /* PhaseFunction::SetG, medium.cpp:31-39 */
        if (g > 1 || g < -1) { g = g > 1 ? .999 : g; g = g < -1 ? -.999 : g; }
        mi->phase.g = g;
        mi->phase.is_initialized = 1;
    }
    V3 Tr = ExpV(-m.sigma_t * Min(t, FLT_MAX) * Length(ray.d));
    V3 density = sampledMedium ? (m.sigma_t * Tr) : Tr;
    Float pdf = 0;
    for (int i = 0; i < nn; ++i) pdf += density[i];
    pdf *= 1 / (Float)nn;
    if (IsZero(pdf)) pdf = 1;
    return sampledMedium ? (Tr * m.sigma_s / pdf) : (Tr / pdf);
}

/* ---- primitive.cpp:246-339 ------------------------------------------------------------------------------ */
static V3 EstimateDirect(const Scene &sc, const Interaction &it, const BSDF *bsdf, const PhaseFunction *phase, const V2 &uScattering,
                         int lightIndex, const V2 &uLight, bool handleMedium) {
    const Light &light = sc.lights[lightIndex];
    int bsdfFlags = BSDF_ALL & ~BSDF_SPECULAR;
    V3 Ld(0.f);
    V3 wi;
    Float lightPdf = 0, scatteringPdf = 0;
    VisibilityTester visibility;
    V3 Li = LightSample_Li(sc, light, it, uLight, &wi, &lightPdf, &visibility);
    bool surface = it.IsSurfaceInteraction() && bsdf;
    if (lightPdf > 0 && !IsBlack(Li) && !IsZero(lightPdf)) {
        V3 f;
        if (surface) {
            f = bsdf->f(it.wo, wi, bsdfFlags) * AbsDot(wi, it.n);
            scatteringPdf = bsdf->Pdf(it.wo, wi, bsdfFlags);
        } else {
            Float p = PhaseP(*phase, it.wo, wi);
            f = V3(p);
            scatteringPdf = p;
        }
        if (!IsBlack(f)) {
            if (handleMedium) Li *= VisTr(sc, visibility);
            else if (!Unoccluded(sc, visibility)) Li = V3(0.f);
            if (!IsBlack(Li)) {
                if (IsDeltaLight(light.flags)) Ld += f * Li / lightPdf;
                else {
                    Float weight = PowerHeuristic(1, lightPdf, 1, scatteringPdf);
                    Ld += f * Li * weight / lightPdf;
                }
            }
        }
    }
    if (!IsDeltaLight(light.flags)) {
        V3 f;
        bool sampledSpecular = false;
        if (surface) {
            int sampledType = 0;
            f = bsdf->Sample_f(it.wo, &wi, uScattering, &scatteringPdf, bsdfFlags, &sampledType);
            f *= AbsDot(wi, it.n);
            sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
        } else {
            Float p = PhaseSample_p(*phase, it.wo, &wi, uScattering);
            f = V3(p);
            scatteringPdf = p;
        }
        if (!IsBlack(f) && scatteringPdf > 0 && !IsZero(scatteringPdf)) {
            Float weight = 1;
            if (!sampledSpecular) {
                lightPdf = LightPdf_Li(sc, light, it, wi);
                if (IsZero(lightPdf)) return Ld;
                weight = PowerHeuristic(1, scatteringPdf, 1, lightPdf);
            }
            SurfaceInteraction lightIsect;
            Ray ray = SpawnRay(it, wi);
            V3 Tr(1.f);
            bool found;
            if (handleMedium) found = SceneIntersectTr(sc, ray, &lightIsect, &Tr);
            else { g_ctr.mis_rays++; found = SceneIntersect(sc, ray, &lightIsect); }
            V3 Li2(0.f);
            if (found) {
                if (sc.prims[lightIsect.primitive].light == lightIndex) Li2 = SurfaceLe(sc, lightIsect, -wi);
            } else Li2 = LightLe(sc, light, ray);
            if (!IsBlack(Li2)) Ld += f * Li2 * Tr * weight / scatteringPdf;
        }
    }
    return Ld;
}

/* primitive.cpp:341-358 */
V3 UniformSampleOneLight(const Scene &sc, const Interaction &it, const BSDF *bsdf, const PhaseFunction *phase, V2 u2, V3 u3,
                         bool handleMedium) {
    int nLights = (int)sc.lights.size();
    if (nLights < 1) return V3(0.f);
    int lightNum = (int)Min((Float)(int)(u2[0] * nLights), (Float)(nLights - 1));
    Float lightPdf = Float(1) / nLights;
    V2 uLight(u2[1], u3[0]);
    V2 uScattering(u3[1], u3[2]);
    return EstimateDirect(sc, it, bsdf, phase, uScattering, lightNum, uLight, handleMedium) / lightPdf;
}

/* ---- unit hooks (tests only) ---- */
Float Unit_HG(Float cosTheta, Float g) { return PhaseHG(cosTheta, g); }
V2 Unit_D2_Sample(const EnvMap &e, const V2 &u, Float *pdf) { return D2_SampleContinuous(e, u, pdf); }
Float Unit_D2_Pdf(const EnvMap &e, const V2 &p) { return D2_Pdf(e, p); }

} // namespace orc
