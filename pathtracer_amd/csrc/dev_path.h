/*
 * dev_path.h — per-vertex integrator logic of the wavefront pipeline.
 * Reference: the Render kernel and its integrators (src/main.cpp:72-457), direct lighting
 * (src/core/primitive.cpp:246-358), VisibilityTester (src/core/light.cpp:5-40) and
 * Aggregator::IntersectTr (primitive.cpp:433-461).  The reference runs one thread per pixel
 * through the whole path; here a path is a record in HBM that moves between three kernels per
 * bounce (trace -> shade -> nee).  The arithmetic of one path is kept in the reference's order,
 * including the order in which terms are added into L, so a path's radiance does not depend on
 * how paths are scheduled.
 */
#pragma once
#include "../../include/ptr_api.h"
#include "dev_bssrdf.h"

/* What one vertex hands back to the kernel */
struct VertexOut {
    bool alive;    /* the path continues with the ray written to (ro, rd, medium) */
    bool want_nee; /* the record was filled */
    int nee_flags; /* its flags (NEE_A: shadow ray, NEE_B: MIS ray) */
    bool has_extra; /* SSS only: L_extra is to be added to L after the record's own term */
    f3 L_extra;
    bool defer;     /* SSS_DEFER only: the vertex sampled the transmission of a BSSRDF material: its own part is complete (L, beta with
                       this bounce's factor, flags, RNG position, direct-light record), the tail kernels take it from there */
};
/* shade_vertex's SSS mode: SSS_NONE no BSSRDF in the scene / SSS_INPLACE the BSSRDF tail is traversed in the vertex's lane
 * (k_finish) / SSS_DEFER the vertex ends where its tail would begin and other launches continue it (rt_shade.h).  SSS_BEGIN
 * and SSS_TAIL are roles of those launches (sss_continue below), not modes of shade_vertex. */
enum { SSS_NONE = 0, SSS_INPLACE = 1, SSS_DEFER = 2, SSS_BEGIN = 3, SSS_TAIL = 4 };

/* What a kernel that can take the subsurface branch lends to shade_vertex: the BSSRDF probe
 * segment and the direct light at the exit point are traversed in place (dev_bssrdf.h) */
struct SssCtx {
    const LaneStack *ls;
    NeeCtr *nctr;
    unsigned *probes;
    bool walk; /* VisibilityTester::Tr / IntersectTr may pass through primitives (Li_VolPath, scene not all opaque) */
    ChainRef chain; /* this vertex's probe-chain record (SSS_BEGIN writes it, the tails read it), or {null, -1}: walk in place */
    int slot;       /* SSS_BEGIN: the vertex's path slot, for the record */
    ExitNee exit;   /* SSS_INPLACE: where the exit point's direct light goes if not resolved in place */
};

/* One loop iteration of Li_Direct / Li_Path / Li_PathSampled / Li_VolPath (main.cpp:72-410).
 * On entry (ro, rd) is the ray that was traced and (hp, hf, ht) its closest hit (hp < 0: miss). */
template <int SSS, class NeeSink>
PD VertexOut shade_vertex(const DScene &sc, int integrator, Rng &rng, f3 &ro, f3 &rd, int &ray_medium, int hp, int hf, float ht, f3 &L,
                          f3 &beta, int &bounces, int &pflags, const NeeSink &rec, VertexCtr &ctr, const SssCtx *sss, float *lobe_mem) {
    VertexOut out;
    out.alive = false;
    out.want_nee = false;
    out.nee_flags = 0;
    out.has_extra = false;
    out.L_extra = f3(0.f);
    out.defer = false;
    const bool found = hp >= 0;
    SI si;
    BSDF bsdf;
    bsdf.lm = lobe_mem; /* this lane's column of the kernel's lobe array (dev_bsdf.h) */
    bsdf.n = 0;
    BssrdfParams bp;

    if (integrator == PTR_LI_DIRECT) { /* main.cpp:72-98 */
        ctr.total++;
        if (!found) {
            for (int i = 0; i < sc.n_lights; i++) L += light_le(sc, sc.lights[i], rd);
            return out;
        }
        make_si(sc, hp, hf, ro, rd, ht, ray_medium, &si);
        bsdf_init(&bsdf, si);
        compute_scattering(sc, si, &bsdf, &bp);
        L += surface_le(sc, hp);
        if (sc.n_lights > 0) {
            f2 u2 = rand2(rng);
            f3 u3 = rand3(rng);
            { out.nee_flags = build_nee(sc, ref_from_si(si), &bsdf, 0, u2, u3, false, f3(1.f), false, rec); out.want_nee = out.nee_flags != 0; }
        }
        return out;
    }

    if (integrator == PTR_LI_PATH) { /* main.cpp:362-410 */
        const int max_bounces = 10;
        ctr.total++;
        bool brk = true;
        if (found) {
            make_si(sc, hp, hf, ro, rd, ht, ray_medium, &si);
            bsdf_init(&bsdf, si);
            float pdf = 0.f;
            f2 u = rand2(rng);
            f3 wi, wo = -rd;
            bool scatters = compute_scattering(sc, si, &bsdf, &bp);
            const DPrim &pr = sc.prims[hp];
            f3 Le = (pr.flags & PF_EMISSIVE) ? f3(pr.L[0], pr.L[1], pr.L[2]) : f3(0.f);
            L += beta * Le;
            int st = 0;
            f3 f(0.f);
            if (scatters) f = bsdf_sample_f(bsdf, wo, &wi, u, &pdf, BSDF_ALL, &st);
            if (!is_zero(pdf)) {
                beta *= f * absdot(wi, si.n) / pdf;
                Ref it = ref_from_si(si);
                ro = offset_ray_origin(it.p, it.pError, it.n, wi);
                ray_medium = ref_medium(it, wi);
                rd = wi;
                ctr.surf++;
                brk = false;
                if (max_comp(beta) < 1.f && bounces > 3) {
                    float q = fmaxr((float).05, 1 - max_comp(beta));
                    float uu = rng.next();
                    if (uu < q) brk = true;
                    else beta = beta / (1 - q);
                }
            }
        } else {
            L = sum_lights_le(sc, rd, beta, L);
            ctr.miss++;
        }
        if (brk) {
            if (bounces == max_bounces - 1) L = f3(0.f); /* sic, main.cpp:408 */
            return out;
        }
        bounces++;
        out.alive = bounces < max_bounces;
        return out;
    }

    /* Li_PathSampled (main.cpp:251-354) and Li_VolPath (main.cpp:105-243) */
    const bool vol = integrator == PTR_LI_VOLPATH;
    const int max_bounces = vol ? 15 : 10;
    ctr.total++;
    ctr.miss += !found ? 1 : 0;
    bool scattered_in_medium = false;
    if (vol) {
        bool sampled = false;
        f3 pm;
        float g = 0;
        if (ray_medium >= 0) {
            f2 u = rand2(rng);
            beta *= medium_sample(sc.media[ray_medium], ro, rd, found ? ht : kInfinity, u, &sampled, &pm, &g);
        }
        if (is_black(beta)) return out;
        if (sampled) {
            if (bounces >= max_bounces) return out;
            Ref it;
            it.p = pm; it.pError = f3(0.f); it.n = f3(0.f); it.wo = -rd; it.face = -1; it.med_in = ray_medium; it.med_out = ray_medium;
            f2 u2 = rand2(rng);
            f3 u3 = rand3(rng);
            if (sc.n_lights < 1) ctr.zero++;
            else { out.nee_flags = build_nee(sc, it, nullptr, g, u2, u3, true, beta, true, rec); out.want_nee = out.nee_flags != 0; }
            if (sc.n_lights >= 1 && !out.want_nee) ctr.zero++;
            f3 wo = -rd, wi;
            u2 = rand2(rng);
            phase_sample_p(g, wo, &wi, u2);
            ro = pm; /* OffsetRayOrigin with n = 0 leaves p unchanged */
            rd = wi;
            pflags &= ~PFL_SPECULAR;
            ctr.med++;
            scattered_in_medium = true;
        }
    }
    SH_T(rec.st, SH_MEDIUM);
    if (!scattered_in_medium) {
        if (bounces == 0 || (pflags & PFL_SPECULAR)) {
            if (found) L += beta * surface_le(sc, hp);
            else L = sum_lights_le(sc, rd, beta, L);
        }
        if (!found || bounces >= max_bounces) return out;
        const DPrim &pr = sc.prims[hp];
        if (pr.flags & PF_EMISSIVE) ctr.light++;
        make_si(sc, hp, hf, ro, rd, ht, ray_medium, &si);
        bsdf_init(&bsdf, si);
        SH_T(rec.st, SH_MAKE_SI);
        bool scatters = compute_scattering(sc, si, &bsdf, &bp);
        SH_T(rec.st, SH_SCATTER);
        if (!scatters) return out; /* emitter: its single lobe matches no flag, Sample_f returns pdf 0 */
        Ref it = ref_from_si(si);
        if (bsdf.n == 0) { /* medium boundary without material: pass through, bounce not counted */
            ro = offset_ray_origin(it.p, it.pError, it.n, rd);
            ray_medium = ref_medium(it, rd);
            out.alive = true;
            return out;
        }
        if (bsdf_num_components(bsdf, BSDF_ALL & ~BSDF_SPECULAR) > 0) {
            f2 u2 = rand2(rng);
            f3 u3 = rand3(rng);
            if (sc.n_lights < 1) ctr.zero++;
            else {
                { out.nee_flags = build_nee(sc, it, &bsdf, 0, u2, u3, vol, beta, true, rec); out.want_nee = out.nee_flags != 0; }
                if (!out.want_nee) ctr.zero++;
            }
        }
        float pdf = 0.f;
        f2 u = rand2(rng);
        f3 wi, wo = -rd;
        int flags = 0;
        SH_T(rec.st, SH_NEE_PUT_B);
        f3 f = bsdf_sample_f(bsdf, wo, &wi, u, &pdf, BSDF_ALL, &flags);
        SH_T(rec.st, SH_BSDF_SAMPLE);
        if (is_black(f) || is_zero(pdf)) return out;
        beta *= f * absdot(wi, si.n) / pdf;
        if (flags & BSDF_SPECULAR) pflags |= PFL_SPECULAR;
        else pflags &= ~PFL_SPECULAR;
        ro = offset_ray_origin(it.p, it.pError, it.n, wi);
        ray_medium = ref_medium(it, wi);
        rd = wi;
        if (!vol) ctr.surf++; /* main.cpp:319 counts before the BSSRDF tail, :229 after it */
        if (SSS == SSS_DEFER && bp.has && (flags & BSDF_TRANSMISSION)) {
            out.defer = true; /* this vertex's own part ends here; the tail kernels continue it (sss_continue below) */
            return out;
        }
        if (SSS == SSS_INPLACE && bp.has && (flags & BSDF_TRANSMISSION)) {
            out.has_extra = true;
            if (!bssrdf_step(sc, bsdf, bp, si, rng, vol, sss->walk, &out.L_extra, beta, pflags, ro, rd, ray_medium, *sss->nctr, sss->probes, *sss->ls))
                return out;
        }
        if (vol) ctr.surf++;
    }
    if (max_comp(beta) < 1.f && bounces > 3) {
        float q = fmaxr((float).05, 1 - max_comp(beta));
        float uu = rng.next();
        if (uu < q) return out;
        beta = beta / (1 - q);
    }
    bounces++;
    out.alive = true;
    return out;
}

/* ---- the BSSRDF tail of a staged bounce, in kernels of its own (rt_shade.h) ----------------------------------------
 * k_shade<SSS_DEFER> ends a vertex that sampled the transmission of a BSSRDF material after `beta *= f |cos| / pdf`
 * (main.cpp:199-206): the path record holds L, that beta, the flags and the RNG position, the incoming ray and the hit are
 * left in place.  sss_continue picks the path up there and runs main.cpp:207-240 - the tail, `surface_hits` under Li_VolPath,
 * Russian roulette, the bounce count - in one of three roles:
 *   SSS_BEGIN    set the probe up (entry interaction, material, BSSRDF, three draws) and leave it as the vertex's chain record;
 *   SSS_TAIL     the chain is settled and the exit point's direct light goes to a record: everything is read from the chain
 *                record, nothing is traversed.  settled = false: not such a vertex, left for
 *   SSS_INPLACE  whatever needs a traversal in the lane; without a chain record the probe is set up here as well. */
struct TailOut {
    bool alive, settled, has_extra;
    f3 L_extra;
};
template <int SSS>
PD TailOut sss_continue(const DScene &sc, int integrator, Rng &rng, f3 &ro, f3 &rd, int &ray_medium, int hp, int hf, float ht, f3 &beta, int &bounces,
                        int &pflags, VertexCtr &ctr, const SssCtx *sss, float *lobe_mem) {
    TailOut out;
    out.alive = false;
    out.settled = true;
    out.has_extra = false;
    out.L_extra = f3(0.f);
    const bool vol = integrator == PTR_LI_VOLPATH;
    const bool to_record = sss->exit.planes && sss->chain.valid() && !sss->walk;
    if (SSS == SSS_TAIL && !(chain_settled(sss->chain) && to_record)) {
        out.settled = false;
        return out;
    }
    ProbeSetup ps;
    BSDF bsdf;
    bsdf.lm = lobe_mem;
    bsdf.n = 0;
    if (SSS != SSS_BEGIN && sss->chain.valid()) {
        chain_get_setup(sss->chain, sc, &ps);
        bsdf.ns = ps.b.ns; bsdf.ss = ps.b.ss; bsdf.ts = ps.b.ts;
        rng.dim += BSSRDF_SETUP_DRAWS; /* drawn when the record was made */
    } else {
        SI si;
        BssrdfParams bp;
        make_si(sc, hp, hf, ro, rd, ht, ray_medium, &si);
        bsdf_init(&bsdf, si);
        compute_scattering(sc, si, &bsdf, &bp);
        bssrdf_setup(sc, bsdf, bp, si, rng, &ps);
        if (SSS == SSS_BEGIN) {
            chain_put_setup(sss->chain, sc, ps, sss->slot);
            return out;
        }
    }
    if (!bssrdf_finish<SSS == SSS_INPLACE>(sc, bsdf, ps, rng, vol, sss->walk, &out.L_extra, &out.has_extra, beta, pflags, ro, rd, ray_medium, sss->nctr,
                                           sss->probes, *sss->ls, sss->chain, sss->exit))
        return out;
    if (vol) ctr.surf++; /* main.cpp:229 */
    if (max_comp(beta) < 1.f && bounces > 3) {
        float q = fmaxr((float).05, 1 - max_comp(beta));
        float uu = rng.next();
        if (uu < q) return out;
        beta = beta / (1 - q);
    }
    bounces++;
    out.alive = true;
    return out;
}
