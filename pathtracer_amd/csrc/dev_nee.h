/*
 * dev_nee.h — direct lighting of the wavefront pipeline: the random stream, the path and
 * direct-light records, EstimateDirect split at its two scene queries.
 * Reference: UniformSampleOneLight / EstimateDirect (src/core/primitive.cpp:246-358),
 * VisibilityTester (src/core/light.cpp:5-40), Aggregator::IntersectTr (primitive.cpp:433-461).
 */
#pragma once
#include "../../include/ptr_api.h"
#include "dev_light.h"

/* ---- counter-based RNG (DESIGN.md "RNG"; the reference's cuRAND state seeded with time(0),
 * main.cpp:51,1102, is not reproducible by construction) ---- */
PD uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t key;
    uint32_t dim;
    PD float next() {
        uint64_t z = mix64(key + 0x9E3779B97F4A7C15ULL * ((uint64_t)dim + 1));
        dim++;
        uint32_t x = (uint32_t)(z >> 32);
        float f = (float)x * 2.3283064365386963e-10f;
        return f + 1.1641532182693481e-10f;
    }
};
PD Rng make_rng(uint64_t seed, uint32_t pixel, uint32_t sample) {
    Rng r;
    uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ULL * ((uint64_t)pixel + 1));
    r.key = mix64(k ^ (0xD1B54A32D192ED03ULL * ((uint64_t)sample + 1)));
    r.dim = 0;
    return r;
}
/* Point2f(rand, rand) / Point3f(rand, rand, rand): g++ evaluates arguments right to left */
PD f2 rand2(Rng &r) { float y = r.next(); float x = r.next(); return f2(x, y); }
PD f3 rand3(Rng &r) { float z = r.next(); float y = r.next(); float x = r.next(); return f3(x, y, z); }

/* ---- records in HBM ----
 * A record type of W words is stored as word planes, TILED: slots are grouped in tiles of
 * 2^PLANE_TILE_SHIFT, and a tile holds its W planes back to back (word w of slot s at
 * ((s >> T) * W + w) << T | (s & (2^T - 1))).  Consecutive lanes still read consecutive words, but all
 * the fields of the 64 records a wavefront works on lie within W * 4 KiB instead of W different
 * regions cap * 4 bytes apart - one or two pages for the TLB instead of sixty. */
#define PLANE_TILE_SHIFT 10
PD size_t plane_at(int W, int slot, int w) {
    return ((((size_t)(slot >> PLANE_TILE_SHIFT)) * W + w) << PLANE_TILE_SHIFT) | (size_t)(slot & ((1 << PLANE_TILE_SHIFT) - 1));
}
enum { PFL_SPECULAR = 1 };
/* bits k_shade leaves in Pipe::route for k_route (ptr_runtime.hip); bit q feeds queue q = counts[q] */
enum { ROUTE_NEXT = 1, ROUTE_NEE = 2, ROUTE_SHADOW = 4, ROUTE_MIS = 8 };
struct __attribute__((aligned(16))) PathRec {
    float L[3], beta[3];
    int bounces, flags, dim, medium;
    float weight;
    int pad;
};
enum { NEE_A = 1, NEE_B = 2, NEE_MEDIUM = 4, NEE_COUNT_ZERO = 8, NEE_EXTRA = 16 };
struct __attribute__((aligned(16))) NeeRec {
    float beta[3];
    int flags;
    /* light-sampling half: Ld += fA * LiA * wA / pdfA if the segment is unoccluded */
    float fA[3], LiA[3], wA, pdfA;
    float oA[3], dA[3];
    int medA;
    float p1[3], p1err[3], p1n[3];
    /* BSDF / phase-sampling half: Ld += fB * Le(hit) * Tr * wB / pdfB */
    float fB[3], wB, pdfB;
    float oB[3], dB[3];
    int medB, light;
    /* results of the two queued scene queries (persistent tracer), read by the resolve kernel */
    int occA, hitB;
    float tB;
    float tA; /* shadow ray traced as a closest-hit ray (occA = primitive or -1): where it ended (SHADOW_CLOSEST below) */
    /* NEE_EXTRA: a term added to L right after this record's (direct light at a BSSRDF exit point) */
    float extra[3];
};

PD void st3(float *d, const f3 &v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
PD f3 ld3(const float *s) { return f3(s[0], s[1], s[2]); }

/* Where build_nee leaves the record it makes: NeeToStruct fills a NeeRec in registers (finish
 * kernel), NeeToPlanes stores each word straight into its plane in HBM (shade kernel: word w of
 * slot s at base[w * cap + s], a coalesced store per field and nothing kept in registers). */
#define NEE_W(field) ((int)(offsetof(NeeRec, field) / 4))
/* -DSH_STATS (diagnostic build, tools/gpu_shadestats.py): wave cycles per section of a shaded vertex */
#ifdef SH_STATS
enum { SH_LOAD = 0, SH_MEDIUM, SH_MAKE_SI, SH_SCATTER, SH_LIGHT_SAMPLE, SH_BSDF_EVAL, SH_NEE_PUT_A, SH_BSDF_SAMPLE_MIS, SH_LIGHT_PDF, SH_NEE_PUT_B,
       SH_BSDF_SAMPLE, SH_TAIL, SH_STORE, SH_VERTS, SH_N };
struct ShStats { long long last; unsigned long long acc[SH_N]; };
static __device__ unsigned long long g_sh_stats[SH_N];
#define SH_T(st, k) do { if (!(st)) break; const long long now_ = clock64(); (st)->acc[k] += (unsigned long long)(now_ - (st)->last); (st)->last = now_; } while (0)
#define SH_STATS_PARAM , ShStats *shs
#define SH_STATS_ARG , shs
#else
#define SH_T(st, k) do { } while (0)
#define SH_STATS_PARAM
#define SH_STATS_ARG
#endif
struct NeeToStruct {
    NeeRec *r;
#ifdef SH_STATS
    ShStats *st = nullptr;
#endif
    PD void put(int w, float v) const { reinterpret_cast<float *>(r)[w] = v; }
    void put(int w, int v) const = delete; /* an index stored through put() would be converted to float: medium 1 -> 0x3f800000 */
    PD void puti(int w, int v) const { reinterpret_cast<int *>(r)[w] = v; }
    PD void put3(int w, const f3 &v) const { put(w, v.x); put(w + 1, v.y); put(w + 2, v.z); }
    PD bool wants_p1() const { return true; }
};
/* The records a vertex leaves behind - its direct-light record (1), its path record (2) - are next read by another kernel, gigabytes
 * of traffic later: stored past the caches (non-temporal), they do not push the scene's tables out of the L2 on their way
 * (k_shade 35.2 -> 34.9 -> 34.6 ms per configs[1] step with 0 -> 1 -> 2; films identical). */
#ifndef PT_NT_RECORDS
#define PT_NT_RECORDS 2
#endif
struct NeeToPlanes {
    float *base;
    int slot;
    bool p1; /* the light point is read only by VisibilityTester::Tr's walk (vis_tr): 9 of the record's words that a pass without walks need not store */
#ifdef SH_STATS
    ShStats *st;
#endif
    PD bool wants_p1() const { return p1; }
#if PT_NT_RECORDS
    PD void put(int w, float v) const { __builtin_nontemporal_store(v, &base[plane_at((int)(sizeof(NeeRec) / 4), slot, w)]); }
#else
    PD void put(int w, float v) const { base[plane_at((int)(sizeof(NeeRec) / 4), slot, w)] = v; }
#endif
    void put(int w, int v) const = delete;
    PD void puti(int w, int v) const { put(w, __int_as_float(v)); }
    PD void put3(int w, const f3 &v) const { put(w, v.x); put(w + 1, v.y); put(w + 2, v.z); }
};

/* k_shade<SSS_BEGIN> replays a vertex only for its probe segment: the direct-light record goes nowhere */
struct NeeToNothing {
#ifdef SH_STATS
    ShStats *st = nullptr;
#endif
    PD void put(int, float) const {}
    void put(int w, int v) const = delete;
    PD void puti(int, int) const {}
    PD void put3(int, const f3 &) const {}
    PD bool wants_p1() const { return false; }
};

struct VertexCtr { unsigned total, miss, surf, med, light, zero; };

/* Interaction::SpawnRayTo, interaction.cpp:11-21 */
PD void spawn_ray_to(const f3 &p, const f3 &pErr, const f3 &n, int med_in, int med_out, const LightPoint &to, f3 *o, f3 *d, int *med) {
    f3 origin = offset_ray_origin(p, pErr, n, to.p - p);
    f3 target = offset_ray_origin(to.p, to.pError, to.n, origin - to.p);
    *o = origin;
    *d = target - origin;
    *med = si_medium(n, med_in, med_out, *d);
}

/* EstimateDirect up to the two scene queries (primitive.cpp:246-339): everything that does not
 * need a ray is evaluated here, the queries themselves are left to the nee kernel.
 * Returns false if neither half has anything to trace. */
template <class NeeSink>
PD int build_nee(const DScene &sc, const Ref &it, const BSDF *bsdf, float phase_g, const f2 &u2, const f3 &u3, bool handleMedium,
                  const f3 &beta, bool countZero, const NeeSink &rec) {
    int nLights = sc.n_lights;
    int lightNum = (int)fminr((float)(int)(u2.x * nLights), (float)(nLights - 1));
    f2 uLight(u2.y, u3.x);
    f2 uScattering(u3.y, u3.z);
    const DLight &light = sc.lights[lightNum];
    const int bsdfFlags = BSDF_ALL & ~BSDF_SPECULAR;
    int flags = 0;
    f3 wi;
    float lightPdf = 0, scatteringPdf = 0;
    LightPoint p1;
    f3 Li = light_sample_li(sc, light, it, uLight, &wi, &lightPdf, &p1);
    SH_T(rec.st, SH_LIGHT_SAMPLE);
    if (lightPdf > 0 && !is_black(Li) && !is_zero(lightPdf)) {
        f3 f;
        if (bsdf) {
            f = bsdf_f_pdf(*bsdf, it.wo, wi, bsdfFlags, &scatteringPdf) * absdot(wi, it.n);
        } else {
            float p = phase_hg(dot(it.wo, wi), phase_g);
            f = f3(p);
            scatteringPdf = p;
        }
        SH_T(rec.st, SH_BSDF_EVAL);
        if (!is_black(f)) {
            flags |= NEE_A;
            rec.put3(NEE_W(fA), f);
            rec.put3(NEE_W(LiA), Li);
            /* a delta light: Ld += f * Li / lightPdf (primitive.cpp:284-285) - the same sum with a weight of exactly 1 */
            rec.put(NEE_W(wA), light_is_delta(light) ? 1.f : power_heuristic(1, lightPdf, 1, scatteringPdf));
            rec.put(NEE_W(pdfA), lightPdf);
            f3 o, d;
            int med;
            spawn_ray_to(it.p, it.pError, it.n, it.med_in, it.med_out, p1, &o, &d, &med);
            rec.put3(NEE_W(oA), o);
            rec.put3(NEE_W(dA), d);
            rec.puti(NEE_W(medA), med);
            if (rec.wants_p1()) {
                rec.put3(NEE_W(p1), p1.p);
                rec.put3(NEE_W(p1err), p1.pError);
                rec.put3(NEE_W(p1n), p1.n);
            }
        }
        SH_T(rec.st, SH_NEE_PUT_A);
    }
    { /* (for a delta light the reference skips this half, primitive.cpp:294: here it is evaluated and dropped - it draws no
       * random numbers and stores nothing unless `ok` - because a branch around it cost every scene 4 % of the shade kernel) */
        f3 f;
        bool sampledSpecular = false;
        if (bsdf) {
            int sampledType = 0;
            f = bsdf_sample_f(*bsdf, it.wo, &wi, uScattering, &scatteringPdf, bsdfFlags, &sampledType);
            f *= absdot(wi, it.n);
            sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
        } else {
            float p = phase_sample_p(phase_g, it.wo, &wi, uScattering);
            f = f3(p);
            scatteringPdf = p;
        }
        SH_T(rec.st, SH_BSDF_SAMPLE_MIS);
        if (!is_black(f) && scatteringPdf > 0 && !is_zero(scatteringPdf)) {
            float weight = 1;
            bool ok = true;
            if (!sampledSpecular) {
                lightPdf = light_pdf_li(sc, light, it, wi);
                if (is_zero(lightPdf)) ok = false;
                else weight = power_heuristic(1, scatteringPdf, 1, lightPdf);
            }
            SH_T(rec.st, SH_LIGHT_PDF);
            if (ok && !light_is_delta(light)) {
                flags |= NEE_B;
                rec.put3(NEE_W(fB), f);
                rec.put(NEE_W(wB), weight);
                rec.put(NEE_W(pdfB), scatteringPdf);
                rec.put3(NEE_W(oB), offset_ray_origin(it.p, it.pError, it.n, wi));
                rec.put3(NEE_W(dB), wi);
                rec.puti(NEE_W(medB), ref_medium(it, wi));
                rec.puti(NEE_W(light), lightNum);
            }
        }
    }
    SH_T(rec.st, SH_NEE_PUT_B);
    if (!(flags & (NEE_A | NEE_B))) return 0;
    if (handleMedium) flags |= NEE_MEDIUM;
    if (countZero) flags |= NEE_COUNT_ZERO;
    rec.puti(NEE_W(flags), flags);
    rec.put3(NEE_W(beta), beta);
    return flags;
}

/* ---- the two scene queries of EstimateDirect and the final sum (primitive.cpp:270-357) ---- */
struct NeeCtr { unsigned shadow, mis, zero; };

/* VisibilityTester::Tr, light.cpp:10-40 */
PD f3 vis_tr(const DScene &sc, const NeeRec &rec, NeeCtr &ctr, const LaneStack &ls) {
    f3 ro = ld3(rec.oA), rd = ld3(rec.dA);
    int med = rec.medA;
    LightPoint p1;
    p1.p = ld3(rec.p1); p1.pError = ld3(rec.p1err); p1.n = ld3(rec.p1n);
    f3 Tr(1.f);
    for (int guard = 0; guard < 256; guard++) {
        float tmax = 1 - kShadowEpsilon;
        int hp = -1, hf = 0;
        float b0, b1;
        ctr.shadow++;
        bool hit = traverse<false, false>(sc, ro, rd, tmax, hp, hf, b0, b1, nullptr, ls);
        if (hit && sc.prims[hp].material >= 0) return f3(0.0f);
        if (med >= 0) Tr *= medium_tr(sc.media[med], rd, tmax);
        if (!hit || (sc.prims[hp].flags & PF_EMISSIVE)) break;
        SI si;
        make_si(sc, hp, hf, ro, rd, tmax, med, &si);
        spawn_ray_to(si.p, si.pError, si.n, si.med_in, si.med_out, p1, &ro, &rd, &med);
    }
    return Tr;
}
/* Aggregator::IntersectTr, primitive.cpp:433-461 */
PD bool intersect_tr(const DScene &sc, f3 ro, const f3 &rd, int med, int *hprim, f3 *Tr, NeeCtr &ctr, const LaneStack &ls) {
    *Tr = f3(1.f);
    for (int guard = 0; guard < 256; guard++) {
        float tmax = kInfinity;
        int hp = -1, hf = 0;
        float b0, b1;
        ctr.mis++;
        bool hit = traverse<false, false>(sc, ro, rd, tmax, hp, hf, b0, b1, nullptr, ls);
        if (med >= 0) *Tr *= medium_tr(sc.media[med], rd, tmax);
        if (!hit || (sc.prims[hp].flags & PF_EMISSIVE)) return false;
        if (sc.prims[hp].material >= 0) { *hprim = hp; return true; }
        SI si;
        make_si(sc, hp, hf, ro, rd, tmax, med, &si);
        ro = offset_ray_origin(si.p, si.pError, si.n, rd);
        med = si_medium(si.n, si.med_in, si.med_out, rd);
    }
    return false;
}

/* Stage 1 of a direct-light record: the visibility / transmittance query of the light-sampling
 * half (primitive.cpp:283-290).  Leaves Li·Tr (or 0) in rec->LiA.
 * LOOP = false is valid whenever the walk of VisibilityTester::Tr cannot go past its first hit:
 * outside Li_VolPath (Unoccluded, light.cpp:5-8), or when every primitive has a material - then
 * "any hit" decides, and the medium term is that of the unbroken segment. */
template <bool LOOP>
PD void nee_shadow(const DScene &sc, NeeRec *rec, NeeCtr &ctr, const LaneStack &ls) {
    f3 Li = ld3(rec->LiA);
    const bool medium = (rec->flags & NEE_MEDIUM) != 0;
    if (LOOP) Li *= vis_tr(sc, *rec, ctr, ls);
    else {
        float tmax = 1 - kShadowEpsilon;
        int hp = -1, hf = 0;
        float b0, b1;
        ctr.shadow++;
        f3 rd = ld3(rec->dA);
        if (traverse<true, false>(sc, ld3(rec->oA), rd, tmax, hp, hf, b0, b1, nullptr, ls)) Li = f3(0.f);
        else if (medium) {
            f3 Tr(1.f);
            if (rec->medA >= 0) Tr *= medium_tr(sc.media[rec->medA], rd, tmax);
            Li *= Tr;
        }
    }
    st3(rec->LiA, Li);
}

/* Stage 2: the query of the BSDF / phase-sampling half (primitive.cpp:315-336) and the sum */
template <bool LOOP>
PD f3 nee_mis_resolve(const DScene &sc, const NeeRec &rec, NeeCtr &ctr, const LaneStack &ls) {
    f3 Ld(0.f);
    const bool medium = (rec.flags & NEE_MEDIUM) != 0;
    if (rec.flags & NEE_A) {
        f3 Li = ld3(rec.LiA);
        if (!is_black(Li)) Ld += ld3(rec.fA) * Li * rec.wA / rec.pdfA;
    }
    if (rec.flags & NEE_B) {
        f3 Tr(1.f);
        bool found;
        int hp = -1;
        f3 rd = ld3(rec.dB);
        if (LOOP) found = intersect_tr(sc, ld3(rec.oB), rd, rec.medB, &hp, &Tr, ctr, ls);
        else {
            float tmax = kInfinity;
            int hf = 0;
            float b0, b1;
            ctr.mis++;
            found = traverse<false, false>(sc, ld3(rec.oB), rd, tmax, hp, hf, b0, b1, nullptr, ls);
            if (medium) {
                if (rec.medB >= 0) Tr *= medium_tr(sc.media[rec.medB], rd, tmax);
                if (found && (sc.prims[hp].flags & PF_EMISSIVE)) found = false; /* IntersectTr: an emitter ends the walk with "false" */
            }
        }
        f3 Li2(0.f);
        if (found) {
            if (sc.prims[hp].light == rec.light) Li2 = surface_le(sc, hp);
        } else Li2 = light_le(sc, sc.lights[rec.light], rd);
        if (!is_black(Li2)) Ld += ld3(rec.fB) * Li2 * Tr * rec.wB / rec.pdfB;
    }
    /* UniformSampleOneLight: / lightPdf with lightPdf = 1 / nLights, primitive.cpp:341-358 */
    float selPdf = float(1) / sc.n_lights;
    f3 uL = Ld / selPdf;
    f3 out = ld3(rec.beta) * uL;
    if ((rec.flags & NEE_COUNT_ZERO) && is_black(out)) ctr.zero++;
    return out;
}

/* Both stages for the case where the walks cannot go past their first hit (see nee_shadow): the
 * any-hit result of the shadow ray (rec.occA) and the closest hit of the MIS ray (rec.hitB, rec.tB)
 * were produced by the persistent tracer. */
/* shadow_closest: Li_VolPath in a scene whose only primitives without a material are emitters.  VisibilityTester::Tr
 * (light.cpp:10-40) then ends at its first hit whatever it is - a material occludes, an emitter ends the walk with the
 * transmittance up to it - so it is ONE closest-hit query of the persistent tracer: occA = the primitive (or -1), tA = where
 * the segment ended (the hit, or 1 - ShadowEpsilon). */
PD f3 nee_resolve_traced(const DScene &sc, const NeeRec &rec, NeeCtr &ctr, bool shadow_closest) {
    f3 Ld(0.f);
    const bool medium = (rec.flags & NEE_MEDIUM) != 0;
    if (rec.flags & NEE_A) {
        f3 Li = ld3(rec.LiA);
        ctr.shadow++;
        if (shadow_closest) {
            if (rec.occA >= 0 && sc.prims[rec.occA].material >= 0) Li = f3(0.f);
            else {
                f3 Tr(1.f);
                if (rec.medA >= 0) Tr *= medium_tr(sc.media[rec.medA], ld3(rec.dA), rec.tA);
                Li *= Tr;
            }
        } else if (rec.occA) Li = f3(0.f);
        else if (medium) {
            f3 Tr(1.f);
            if (rec.medA >= 0) Tr *= medium_tr(sc.media[rec.medA], ld3(rec.dA), 1 - kShadowEpsilon);
            Li *= Tr;
        }
        if (!is_black(Li)) Ld += ld3(rec.fA) * Li * rec.wA / rec.pdfA;
    }
    if (rec.flags & NEE_B) {
        f3 Tr(1.f);
        f3 rd = ld3(rec.dB);
        int hp = rec.hitB;
        bool found = hp >= 0;
        ctr.mis++;
        if (medium) {
            if (rec.medB >= 0) Tr *= medium_tr(sc.media[rec.medB], rd, rec.tB);
            if (found && (sc.prims[hp].flags & PF_EMISSIVE)) found = false;
        }
        f3 Li2(0.f);
        if (found) {
            if (sc.prims[hp].light == rec.light) Li2 = surface_le(sc, hp);
        } else Li2 = light_le(sc, sc.lights[rec.light], rd);
        if (!is_black(Li2)) Ld += ld3(rec.fB) * Li2 * Tr * rec.wB / rec.pdfB;
    }
    float selPdf = float(1) / sc.n_lights;
    f3 uL = Ld / selPdf;
    f3 out = ld3(rec.beta) * uL;
    if ((rec.flags & NEE_COUNT_ZERO) && is_black(out)) ctr.zero++;
    return out;
}
