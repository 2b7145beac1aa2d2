/*
 * orc_render.cpp — ORACLE (test infrastructure only).  PARITY UNPINNED.
 * Restates the render kernel and integrators of src/main.cpp:24-457 and Camera::SpawnRay
 * (src/core/camera.cpp:219-224).
 *
 * RNG: the reference uses cuRAND XORWOW seeded with time(0) (main.cpp:51,1102); cuRAND is not
 * available here and the seed is never reproducible, so both the oracle and the HIP path use
 * the counter-based generator specified in DESIGN.md ("RNG"): SplitMix64 keyed by
 * (seed, pixel, sample), output mapped to (0,1] with cuRAND's published uniform formula.
 * Draw ORDER follows g++'s right-to-left evaluation of constructor arguments, e.g.
 * Point2f u(rand_float(s), rand_float(s)) draws u.y first (SURVEY §7 "evaluation order").
 */
#include "orc_scene.h"

namespace orc {

static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
RNG MakeRNG(uint64_t seed, uint32_t pixel, uint32_t sample) {
    RNG r;
    uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ULL * ((uint64_t)pixel + 1));
    r.key = mix64(k ^ (0xD1B54A32D192ED03ULL * ((uint64_t)sample + 1)));
    r.dim = 0;
    return r;
}
Float RNG::next() {
    uint64_t z = mix64(key + 0x9E3779B97F4A7C15ULL * ((uint64_t)dim + 1));
    dim++;
    uint32_t x = (uint32_t)(z >> 32);
    /* curand_uniform: x * 2^-32 + 2^-33 in float arithmetic -> (0, 1] */
    Float f = (Float)x * 2.3283064365386963e-10f;
    return f + 1.1641532182693481e-10f;
}
static inline V2 rand2(RNG &r) { Float y = r.next(); Float x = r.next(); return V2(x, y); }
static inline V3 rand3(RNG &r) { Float z = r.next(); Float y = r.next(); Float x = r.next(); return V3(x, y, z); }

/* camera.cpp:219-224 */
Ray CameraSpawnRay(const Camera &c, Float s, Float t, const V2 &disk, int medium) {
    V2 rd = c.lensRadius * disk;
    V3 offset = c.u * rd.x + c.v * rd.y;
    V3 d = c.lower_left + s * c.horizontal + t * c.vertical - c.position - offset;
    return Ray(c.position + offset, Normalize(d), ORC_Infinity, 0, medium);
}

/* main.cpp:412-419 */
Float GetFilterWeight(Float u0, Float u1, Float u, Float v) {
    Float x = Absf(u - u0);
    Float y = Absf(v - u1);
    Float alpha = 2.f;
    Float e = expf(-alpha);
    return Max(0.f, Float(expf(-alpha * x * x) - e)) * Max(0.f, Float(expf(-alpha * y * y) - e));
}

static V3 SumLightsLe(const Scene &sc, const Ray &ray, const V3 &beta, V3 L) {
    for (size_t i = 0; i < sc.lights.size(); i++) L += beta * LightLe(sc, sc.lights[i], ray);
    return L;
}

/* the BSSRDF tail shared by Li_PathSampled (main.cpp:320-342) and Li_VolPath (:207-228) */
static bool BSSRDFStep(const Scene &sc, BSDF &bsdf, const SurfaceInteraction &isect, RNG &rng, bool handleMedium, V3 &L, V3 &beta,
                       bool &specularBounce, Ray &ray) {
    (void)isect;
    SurfaceInteraction pi;
    Float pdf = 0;
    bsdf.nBxDFs = 0;
    Float u1 = rng.next();
    V2 u2 = rand2(rng);
    V3 S = BSSRDF_Sample_S(sc, bsdf.bssrdf, &bsdf, u1, u2, &pi, &pdf);
    if (IsBlack(S) || IsZero(pdf)) return false;
    beta *= S / pdf;
    u2 = rand2(rng);
    V3 u3 = rand3(rng);
    V3 uL = UniformSampleOneLight(sc, pi, &bsdf, nullptr, u2, u3, handleMedium);
    L += beta * uL;
    u2 = rand2(rng);
    V3 wi;
    int flags = 0;
    V3 f = bsdf.Sample_f(pi.wo, &wi, u2, &pdf, BSDF_ALL, &flags);
    if (IsBlack(f) || IsZero(pdf)) return false;
    beta *= f * AbsDot(wi, pi.n) / pdf;
    specularBounce = (flags & BSDF_SPECULAR) != 0;
    ray = SpawnRay(pi, wi);
    return true;
}

/* main.cpp:251-354 */
static V3 Li_PathSampled(const Scene &sc, const Ray &r, RNG &rng) {
    V3 L(0.f), beta(1.f);
    Ray ray(r);
    bool specularBounce = false;
    int bounces;
    const int max_bounces = 10;
    const Float rrThreshold = 1;
    for (bounces = 0;; bounces++) {
        SurfaceInteraction isect;
        bool found = SceneIntersect(sc, ray, &isect);
        g_ctr.total_paths++;
        g_ctr.misses += !found ? 1 : 0;
        if (bounces == 0 || specularBounce) {
            if (found) L += beta * SurfaceLe(sc, isect, -ray.d);
            else L = SumLightsLe(sc, ray, beta, L);
        }
        if (!found || bounces >= max_bounces) break;
        if (sc.prims[isect.primitive].emissive) g_ctr.light_hits++;
        BSDF bsdf(isect);
        ComputeScatteringFunctions(sc, &isect, &bsdf);
        if (bsdf.nBxDFs == 0) {
            ray = SpawnRay(isect, ray.d);
            bounces--;
            continue;
        }
        if (bsdf.NumComponents(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
            V2 u2 = rand2(rng);
            V3 u3 = rand3(rng);
            V3 Ld = beta * UniformSampleOneLight(sc, isect, &bsdf, nullptr, u2, u3, false);
            if (IsBlack(Ld)) g_ctr.zero_radiance++;
            L += Ld;
        }
        Float pdf = 0.f;
        V2 u = rand2(rng);
        V3 wi, wo = -ray.d;
        int flags = 0;
        V3 f = bsdf.Sample_f(wo, &wi, u, &pdf, BSDF_ALL, &flags);
        if (IsBlack(f) || IsZero(pdf)) break;
        beta *= f * AbsDot(wi, isect.n) / pdf;
        specularBounce = (flags & BSDF_SPECULAR) != 0;
        ray = SpawnRay(isect, wi);
        g_ctr.surface_hits++;
        if (bsdf.has_bssrdf && (flags & BSDF_TRANSMISSION)) {
            if (!BSSRDFStep(sc, bsdf, isect, rng, false, L, beta, specularBounce, ray)) break;
        }
        V3 rrBeta = beta;
        if (MaxComponent(rrBeta) < rrThreshold && bounces > 3) {
            Float q = Max((Float).05, 1 - MaxComponent(rrBeta));
            Float u = rng.next();
            if (u < q) break;
            beta = beta / (1 - q);
        }
    }
    return L;
}

/* main.cpp:105-243 */
static V3 Li_VolPath(const Scene &sc, const Ray &r, RNG &rng) {
    V3 L(0.f), beta(1.f);
    Ray ray(r);
    bool specularBounce = false;
    int bounces;
    const int max_bounces = 15;
    const Float rrThreshold = 1;
    for (bounces = 0;; bounces++) {
        SurfaceInteraction isect;
        bool found = SceneIntersect(sc, ray, &isect);
        g_ctr.total_paths++;
        g_ctr.misses += !found ? 1 : 0;
        MediumInteraction mi;
        if (ray.medium >= 0) {
            V2 u = rand2(rng);
            beta *= MediumSample(sc, ray.medium, ray, u, &mi);
        }
        if (IsBlack(beta)) break;
        if (mi.IsValid()) {
            if (bounces >= max_bounces) break;
            V2 u2 = rand2(rng);
            V3 u3 = rand3(rng);
            V3 Ld = beta * UniformSampleOneLight(sc, mi, nullptr, &mi.phase, u2, u3, true);
            if (IsBlack(Ld)) g_ctr.zero_radiance++;
            L += Ld;
            V3 wo = -ray.d, wi;
            u2 = rand2(rng);
            PhaseSample_p(mi.phase, wo, &wi, u2);
            ray = SpawnRay(mi, wi);
            specularBounce = false;
            g_ctr.medium_hits++;
        } else {
            if (bounces == 0 || specularBounce) {
                if (found) L += beta * SurfaceLe(sc, isect, -ray.d);
                else L = SumLightsLe(sc, ray, beta, L);
            }
            if (!found || bounces >= max_bounces) break;
            if (sc.prims[isect.primitive].emissive) g_ctr.light_hits++;
            BSDF bsdf(isect);
            ComputeScatteringFunctions(sc, &isect, &bsdf);
            if (bsdf.nBxDFs == 0) {
                ray = SpawnRay(isect, ray.d);
                bounces--;
                continue;
            }
            if (bsdf.NumComponents(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
                V2 u2 = rand2(rng);
                V3 u3 = rand3(rng);
                V3 Ld = beta * UniformSampleOneLight(sc, isect, &bsdf, nullptr, u2, u3, true);
                if (IsBlack(Ld)) g_ctr.zero_radiance++;
                L += Ld;
            }
            Float pdf = 0.f;
            V2 u = rand2(rng);
            V3 wi, wo = -ray.d;
            int flags = 0;
            V3 f = bsdf.Sample_f(wo, &wi, u, &pdf, BSDF_ALL, &flags);
            if (IsBlack(f) || IsZero(pdf)) break;
            beta *= f * AbsDot(wi, isect.n) / pdf;
            specularBounce = (flags & BSDF_SPECULAR) != 0;
            ray = SpawnRay(isect, wi);
            if (bsdf.has_bssrdf && (flags & BSDF_TRANSMISSION)) {
                if (!BSSRDFStep(sc, bsdf, isect, rng, true, L, beta, specularBounce, ray)) break;
            }
            g_ctr.surface_hits++;
        }
        V3 rrBeta = beta;
        if (MaxComponent(rrBeta) < rrThreshold && bounces > 3) {
            Float q = Max((Float).05, 1 - MaxComponent(rrBeta));
            Float u = rng.next();
            if (u < q) break;
            beta = beta / (1 - q);
        }
    }
    return L;
}

/* main.cpp:362-410 */
static V3 Li_Path(const Scene &sc, const Ray &r, RNG &rng) {
    V3 L(0.f), beta(1.f);
    Ray ray(r);
    const Float rrThreshold = 1;
    const int max_bounces = 10;
    int bounces = 0;
    for (bounces = 0; bounces < max_bounces; bounces++) {
        SurfaceInteraction isect;
        g_ctr.total_paths++;
        if (SceneIntersect(sc, ray, &isect)) {
            BSDF bsdf(isect);
            Float pdf = 0.f;
            V2 u = rand2(rng);
            V3 wi, wo = -ray.d;
            ComputeScatteringFunctions(sc, &isect, &bsdf);
            V3 Le = sc.prims[isect.primitive].emissive ? sc.prims[isect.primitive].L : V3(0.f); /* Primitive::Le() */
            L += beta * Le;
            V3 f = bsdf.Sample_f(wo, &wi, u, &pdf, BSDF_ALL);
            if (IsZero(pdf)) break;
            beta *= f * AbsDot(wi, isect.n) / pdf;
            ray = SpawnRay(isect, wi);
            g_ctr.surface_hits++;
            V3 rrBeta = beta;
            if (MaxComponent(rrBeta) < rrThreshold && bounces > 3) {
                Float q = Max((Float).05, 1 - MaxComponent(rrBeta));
                Float uu = rng.next();
                if (uu < q) break;
                beta = beta / (1 - q);
            }
        } else {
            L = SumLightsLe(sc, ray, beta, L);
            g_ctr.misses++;
            break;
        }
    }
    if (bounces == max_bounces - 1) return V3(0.f); /* sic */
    return L;
}

/* main.cpp:72-98 */
static V3 Li_Direct(const Scene &sc, const Ray &ray, RNG &rng) {
    V3 L(0.f);
    SurfaceInteraction isect;
    g_ctr.total_paths++;
    if (!SceneIntersect(sc, ray, &isect)) {
        for (size_t i = 0; i < sc.lights.size(); i++) L += LightLe(sc, sc.lights[i], ray);
    } else {
        BSDF bsdf(isect);
        ComputeScatteringFunctions(sc, &isect, &bsdf);
        V3 wo = isect.wo;
        L += SurfaceLe(sc, isect, wo);
        if (sc.lights.size() > 0) {
            V2 u2 = rand2(rng);
            V3 u3 = rand3(rng);
            L += UniformSampleOneLight(sc, isect, &bsdf, nullptr, u2, u3, false);
        }
    }
    return L;
}

V3 Li(const Scene &sc, int integrator, const Ray &ray, RNG &rng) {
    switch (integrator) {
    case 0: return Li_Direct(sc, ray, rng);
    case 1: return Li_Path(sc, ray, rng);
    case 2: return Li_PathSampled(sc, ray, rng);
    default: return Li_VolPath(sc, ray, rng);
    }
}

} // namespace orc
