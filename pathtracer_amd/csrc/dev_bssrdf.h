/*
 * dev_bssrdf.h — tabulated separable BSSRDF on the device (Subsurface / KdSubsurface materials).
 * Reference: SeparableBSSRDF src/core/bssrdf.h:84-160 and bssrdf.cpp:383-581 (Sr, Pdf_Sr,
 * Sample_Sr, Sample_Sp, Pdf_Sp, Sample_S), Catmull-Rom utilities bssrdf.cpp:32-191,328-357,
 * SubsurfaceFromDiffuse :359-369, and the BSSRDF tail of the integrators (main.cpp:207-228,
 * 320-342).  The profile table itself is built on the host at upload (ptr_tables.cpp).
 *
 * The probe segment of Sample_Sp is the one place on the path where a vertex issues an unknown
 * number of scene queries before it can go on, so it is traversed in-kernel (traverse<> on the
 * lane's own stack) instead of through the ray queues.  The reference keeps up to 256 surface
 * interactions in a 39 KB per-thread array and picks one afterwards; here the chain is walked
 * twice - once to count the hits on the probed primitive, once more up to the chosen one - which
 * needs no per-lane storage and visits the same hits (the chain is deterministic).
 */
#pragma once
#include "dev_nee.h"

struct Bssrdf {
    f3 po_p;
    int po_prim;
    f3 ns, ss, ts; /* frame at the point of entry (bssrdf.h:108-117) */
    float eta;
    f3 sigma_t, rho;
    const DTable *t;
};

PD float comp3(const f3 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

/* geometry.h:1091-1107 over a node array with pred nodes[i] <= x */
PD int interval_leq(const float *nodes, int size, float x) {
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, mid = first + half;
        if (nodes[mid] <= x) { first = mid + 1; len -= half + 1; }
        else len = half;
    }
    return clampi(first - 1, 0, size - 2);
}

/* CatmullRomWeights, bssrdf.cpp:32-67 */
PD bool spline_weights(int size, const float *nodes, float x, int *offset, float &w0, float &w1, float &w2, float &w3) {
    if (!(x >= nodes[0] && x <= nodes[size - 1])) return false;
    int idx = interval_leq(nodes, size, x);
    *offset = idx - 1;
    float x0 = nodes[idx], x1 = nodes[idx + 1];
    float t = (x - x0) / (x1 - x0), t2 = t * t, t3 = t2 * t;
    w1 = 2 * t3 - 3 * t2 + 1;
    w2 = -2 * t3 + 3 * t2;
    if (idx > 0) {
        float a = (t3 - 2 * t2 + t) * (x1 - x0) / (x1 - nodes[idx - 1]);
        w0 = -a;
        w2 += a;
    } else {
        float a = t3 - 2 * t2 + t;
        w0 = 0;
        w1 -= a;
        w2 += a;
    }
    if (idx + 2 < size) {
        float c = (t3 - t2) * (x1 - x0) / (nodes[idx + 2] - x0);
        w1 -= c;
        w3 = c;
    } else {
        float c = t3 - t2;
        w1 -= c;
        w2 += c;
        w3 = 0;
    }
    return true;
}

/* one column of a (size1 x size2) table blended over 4 rows; rows with weight 0 are not read
 * (the first / last row offset may lie outside the table) */
struct RowBlend {
    const float *base; /* array + offset * size2 */
    int size2;
    float w0, w1, w2, w3;
    PD float at(const float *array_at_offset, int idx) const {
        float v = 0;
        if (w0 != 0) v += array_at_offset[idx] * w0;
        if (w1 != 0) v += array_at_offset[size2 + idx] * w1;
        if (w2 != 0) v += array_at_offset[2 * size2 + idx] * w2;
        if (w3 != 0) v += array_at_offset[3 * size2 + idx] * w3;
        return v;
    }
};

/* SampleCatmullRom2D, bssrdf.cpp:69-139 */
PD float spline_sample_2d(int size1, int size2, const float *nodes1, const float *nodes2, const float *values, const float *cdf, float alpha,
                          float u) {
    int offset;
    RowBlend rb;
    rb.size2 = size2;
    if (!spline_weights(size1, nodes1, alpha, &offset, rb.w0, rb.w1, rb.w2, rb.w3)) return 0;
    const float *cdf_o = cdf + (long)offset * size2, *val_o = values + (long)offset * size2;
    float maximum = rb.at(cdf_o, size2 - 1);
    u *= maximum;
    int idx;
    {
        int first = 0, len = size2;
        while (len > 0) {
            int half = len >> 1, mid = first + half;
            if (rb.at(cdf_o, mid) <= u) { first = mid + 1; len -= half + 1; }
            else len = half;
        }
        idx = clampi(first - 1, 0, size2 - 2);
    }
    float f0 = rb.at(val_o, idx), f1 = rb.at(val_o, idx + 1);
    float x0 = nodes2[idx], x1 = nodes2[idx + 1];
    float width = x1 - x0;
    float d0, d1;
    u = (u - rb.at(cdf_o, idx)) / width;
    if (idx > 0) d0 = width * (f1 - rb.at(val_o, idx - 1)) / (x1 - nodes2[idx - 1]);
    else d0 = f1 - f0;
    if (idx + 2 < size2) d1 = width * (rb.at(val_o, idx + 2) - f0) / (nodes2[idx + 2] - x0);
    else d1 = f1 - f0;
    float t;
    if (f0 != f1) t = (f0 - sqrtf(fmaxr((float)0, f0 * f0 + 2 * u * (f1 - f0)))) / (f0 - f1);
    else t = u / f0;
    float a = 0, b = 1, Fhat, fhat;
    for (int guard = 0; guard < 128; guard++) { /* the bracket halves whenever Newton leaves it: ends in ~25 rounds */
        if (!(t >= a && t <= b)) t = 0.5f * (a + b);
        Fhat = t * (f0 + t * (.5f * d0 + t * ((1.f / 3.f) * (-2 * d0 - d1) + f1 - f0 + t * (.25f * (d0 + d1) + .5f * (f0 - f1)))));
        fhat = f0 + t * (d0 + t * (-2 * d0 - d1 + 3 * (f1 - f0) + t * (d0 + d1 + 2 * (f0 - f1))));
        if (absr(Fhat - u) < 1e-6f || b - a < 1e-6f) break;
        if (Fhat - u < 0) a = t;
        else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + width * t;
}

/* InvertCatmullRom, bssrdf.cpp:328-357 */
PD float spline_invert(int n, const float *x, const float *values, float u) {
    if (!(u > values[0])) return x[0];
    else if (!(u < values[n - 1])) return x[n - 1];
    int i = interval_leq(values, n, u);
    float x0 = x[i], x1 = x[i + 1];
    float f0 = values[i], f1 = values[i + 1];
    float width = x1 - x0;
    float d0, d1;
    if (i > 0) d0 = width * (f1 - values[i - 1]) / (x1 - x[i - 1]);
    else d0 = f1 - f0;
    if (i + 2 < n) d1 = width * (values[i + 2] - f0) / (x[i + 2] - x0);
    else d1 = f1 - f0;
    float a = 0, b = 1, t = .5f;
    float Fhat, fhat;
    for (int guard = 0; guard < 128; guard++) {
        if (!(t > a && t < b)) t = 0.5f * (a + b);
        float t2 = t * t, t3 = t2 * t;
        Fhat = (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
        fhat = (6 * t2 - 6 * t) * f0 + (-6 * t2 + 6 * t) * f1 + (3 * t2 - 4 * t + 1) * d0 + (3 * t2 - 2 * t) * d1;
        if (absr(Fhat - u) < 1e-6f || b - a < 1e-6f) break;
        if (Fhat - u < 0) a = t;
        else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + t * width;
}

/* the material's BSSRDF at a vertex: subsurface.cpp:60-73 / kdsubsurface.cpp:78-93 (coefficients,
 * KdSubsurface inverts the diffuse reflectance first: SubsurfaceFromDiffuse) + bssrdf.cpp:726-735 */
PD void bssrdf_make(const DScene &sc, const BssrdfParams &bp, const SI &po, const BSDF &frame, Bssrdf *b) {
    const DTable *t = &sc.tables[bp.table];
    f3 sa = bp.sigma_a, ss = bp.sigma_s;
    if (bp.kd_mode) {
        float r0 = spline_invert(t->nRho, t->rhoSamples, t->rhoEff, bp.kd.x);
        float r1 = spline_invert(t->nRho, t->rhoSamples, t->rhoEff, bp.kd.y);
        float r2 = spline_invert(t->nRho, t->rhoSamples, t->rhoEff, bp.kd.z);
        ss = f3(r0 / bp.mfree.x, r1 / bp.mfree.y, r2 / bp.mfree.z);
        sa = f3((1 - r0) / bp.mfree.x, (1 - r1) / bp.mfree.y, (1 - r2) / bp.mfree.z);
    }
    b->po_p = po.p;
    b->po_prim = po.prim;
    b->ns = frame.ns; b->ss = frame.ss; b->ts = frame.ts; /* same construction as the BSDF's frame (reflection.h:367-369) */
    b->eta = bp.eta;
    b->t = t;
    b->sigma_t = sa + ss;
    b->rho = f3(!is_zero(b->sigma_t.x) ? ss.x / b->sigma_t.x : 0.f, !is_zero(b->sigma_t.y) ? ss.y / b->sigma_t.y : 0.f,
                !is_zero(b->sigma_t.z) ? ss.z / b->sigma_t.z : 0.f);
}

/* profile(rho row, radius column) blended with both sets of spline weights; skip = which products
 * count as zero (Sr: IsZero(w_rho * w_radius); Pdf_Sr: IsZero of either) */
PD float profile_blend(const DTable *t, int ro, int qo, const float *wr, const float *wq, bool product_rule, float *rhoEff) {
    float sr = 0, re = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!product_rule) {
            if (is_zero(wr[i])) continue;
            re += t->rhoEff[ro + i] * wr[i];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (product_rule) {
                float weight = wr[i] * wq[j];
                if (!is_zero(weight)) sr += weight * t->profile[(ro + i) * t->nRadius + qo + j];
            } else {
                if (is_zero(wq[j])) continue;
                sr += t->profile[(ro + i) * t->nRadius + qo + j] * wr[i] * wq[j];
            }
        }
    }
    if (rhoEff) *rhoEff = re;
    return sr;
}

/* TabulatedBSSRDF::Sr, bssrdf.cpp:383-411 */
PD f3 bssrdf_sr(const Bssrdf &b, float r) {
    const DTable *t = b.t;
    float out[3] = {0, 0, 0};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float rOptical = r * comp3(b.sigma_t, ch);
        int ro, qo;
        float wr[4], wq[4];
        if (!spline_weights(t->nRho, t->rhoSamples, comp3(b.rho, ch), &ro, wr[0], wr[1], wr[2], wr[3]) ||
            !spline_weights(t->nRadius, t->radiusSamples, rOptical, &qo, wq[0], wq[1], wq[2], wq[3]))
            continue;
        float sr = profile_blend(t, ro, qo, wr, wq, true, nullptr);
        if (!is_zero(rOptical)) sr /= 2 * kPi * rOptical;
        out[ch] = sr;
    }
    f3 Sr(out[0], out[1], out[2]);
    Sr *= b.sigma_t * b.sigma_t;
    return clamp11(Sr);
}
/* Pdf_Sr, bssrdf.cpp:413-437 */
PD float bssrdf_pdf_sr(const Bssrdf &b, int ch, float r) {
    const DTable *t = b.t;
    float st = comp3(b.sigma_t, ch);
    float rOptical = r * st;
    int ro, qo;
    float wr[4], wq[4];
    if (!spline_weights(t->nRho, t->rhoSamples, comp3(b.rho, ch), &ro, wr[0], wr[1], wr[2], wr[3]) ||
        !spline_weights(t->nRadius, t->radiusSamples, rOptical, &qo, wq[0], wq[1], wq[2], wq[3]))
        return 0.f;
    float rhoEff = 0;
    float sr = profile_blend(t, ro, qo, wr, wq, false, &rhoEff);
    if (!is_zero(rOptical)) sr /= 2 * kPi * rOptical;
    return fmaxr((float)0, sr * st * st / rhoEff);
}
/* Sample_Sr, bssrdf.cpp:439-445 */
PD float bssrdf_sample_sr(const Bssrdf &b, int ch, float u) {
    float st = comp3(b.sigma_t, ch);
    if (is_zero(st)) return -1;
    const DTable *t = b.t;
    return spline_sample_2d(t->nRho, t->nRadius, t->rhoSamples, t->radiusSamples, t->profile, t->profileCDF, comp3(b.rho, ch), u) / st;
}
/* Pdf_Sp, bssrdf.cpp:562-581 */
PD float bssrdf_pdf_sp(const Bssrdf &b, const f3 &pi_p, const f3 &pi_n) {
    f3 d = b.po_p - pi_p;
    f3 dLocal(dot(b.ss, d), dot(b.ts, d), dot(b.ns, d));
    f3 nLocal(dot(b.ss, pi_n), dot(b.ts, pi_n), dot(b.ns, pi_n));
    float rProj[3] = {sqrtf(dLocal.y * dLocal.y + dLocal.z * dLocal.z), sqrtf(dLocal.z * dLocal.z + dLocal.x * dLocal.x),
                      sqrtf(dLocal.x * dLocal.x + dLocal.y * dLocal.y)};
    float pdf = 0;
    const float axisProb[3] = {.25f, .25f, .5f};
    const float chProb = 1 / (float)3;
#pragma unroll
    for (int axis = 0; axis < 3; ++axis)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) pdf += bssrdf_pdf_sr(b, ch, rProj[axis]) * absr(comp3(nLocal, axis)) * chProb * axisProb[axis];
    return pdf;
}

#define PROBE_MAX_FOUND 256   /* bssrdf.cpp:506: SurfaceInteraction si[256] */
#define PROBE_MAX_STEPS 4096  /* the reference's loop is unbounded; a chain stuck on a degenerate hit ends here */

/* The probe chain of Sample_Sp (bssrdf.cpp:511-546): segments from the last hit towards pTarget
 * (Interaction::SpawnRayTo(Point3f), interaction.cpp:12-16: direction = target - p, not offset).
 * Returns how many hits lay on `prim`; with pick >= 0 stops at that hit and fills *out.
 * The reference keeps all interactions of the chain in a 39 KB per-thread array and indexes it afterwards; here the
 * counting walk remembers what the first PROBE_KEEP hits on `prim` were found with (segment, distance, face, medium:
 * 10 words each) so that the chosen one is rebuilt by make_si from the same inputs - the same interaction, bit for
 * bit - and only a choice beyond PROBE_KEEP walks the chain a second time (round 1 always did). */
#define PROBE_KEEP 4
struct ProbeHit { f3 o, d; float t; int face, med; };
PD int bssrdf_probe_chain(const DScene &sc, const f3 &start, const f3 &pTarget, int prim, int pick, SI *out, unsigned *probes,
                          const LaneStack &ls, ProbeHit *keep = nullptr) {
    f3 p = start, pErr(0.f), n(0.f);
    int med_in = -1, med_out = -1;
    int nFound = 0;
    for (int step = 0; step < PROBE_MAX_STEPS; step++) {
        f3 d = pTarget - p;
        f3 o = offset_ray_origin(p, pErr, n, d);
        int med = si_medium(n, med_in, med_out, d);
        float tmax = 1 - kShadowEpsilon, b0, b1;
        int hp = -1, hf = 0;
        if (probes) (*probes)++;
        bool hit = traverse<false, false>(sc, o, d, tmax, hp, hf, b0, b1, nullptr, ls);
        if (is_black(d) || !hit) break;
        SI si;
        make_si(sc, hp, hf, o, d, tmax, med, &si);
        p = si.p; pErr = si.pError; n = si.n; med_in = si.med_in; med_out = si.med_out;
        if (hp == prim) {
            if (nFound == pick) { *out = si; return nFound + 1; }
            if (keep && nFound < PROBE_KEEP) { keep[nFound].o = o; keep[nFound].d = d; keep[nFound].t = tmax; keep[nFound].face = hf; keep[nFound].med = med; }
            nFound++;
            if (nFound > PROBE_MAX_FOUND - 1) break;
        }
    }
    return nFound;
}

/* ---- the probe chain as rays of the persistent tracer (round 3) ------------------------------------------------------
 * In a staged bounce the chain is not walked in the vertex's lane: k_shade<SSS_BEGIN> sets the probe of a deferred vertex up and
 * leaves a CHAIN RECORD; k_pchain, the persistent tracer with a source that CONTINUES (dev_trace.h ChainSource), traces the
 * chain's current segment and, when a batch of its lanes have an answer, takes each of those chains one hit further
 * (chain_advance: make_si, count / keep, next segment - the body of the loop above) and traces on in the same lane until the
 * chain ends; the tail kernel reads the outcome.  A record is SSS_WORDS 32-bit words in tiled planes (dev_nee.h plane_at)
 * indexed by the chain's number. */
enum {
    SW_O = 0,        /* current segment: origin (3), direction (3) - what the tracer reads */
    SW_D = 3,
    SW_HIT = 6,      /* (unused since the chains are advanced inside the tracer: three spare words) */
    SW_T = 7,
    SW_FACE = 8,
    SW_TARGET = 9,   /* pTarget (3) */
    SW_SLOT = 12,    /* path slot of the vertex */
    SW_STATE = 13,   /* SSS_LIVE while segments remain, then SSS_DONE */
    SW_FOUND = 14,   /* hits on the probed primitive so far */
    SW_STEPS = 15,   /* segments traced so far */
    SW_MED = 16,     /* medium of the current segment */
    SW_PRIM = 17,    /* the probed primitive */
    SW_KEEP = 18,    /* PROBE_KEEP x { o (3), d (3), t, face, med }: what the first hits on it were found with */
    SW_KEEP_WORDS = 9,
    /* what the tail kernels need of the entry vertex (ProbeSetup below), so that they do not shade it again */
    SW_OK = SW_KEEP + PROBE_KEEP * SW_KEEP_WORDS, /* the probe was set up (else S = 0) */
    SW_U1,           /* the first draw, remapped: picks among the hits */
    SW_START,        /* start of the probe segment (3) */
    SW_B_PO = SW_START + 3, /* Bssrdf: po.p (3), frame ns ss ts (9), eta, sigma_t (3), rho (3), table index */
    SW_B_NS = SW_B_PO + 3, SW_B_SS = SW_B_NS + 3, SW_B_TS = SW_B_SS + 3,
    SW_B_ETA = SW_B_TS + 3,
    SW_B_SIGT, SW_B_RHO = SW_B_SIGT + 3,
    SW_B_TABLE = SW_B_RHO + 3,
    SSS_WORDS
};
enum { SSS_LIVE = 1, SSS_DONE = 2 };
struct ChainRef { /* a chain record: planes + number; planes == null: there is none, the chain is walked in place */
    uint32_t *planes;
    int j;
    PD bool valid() const { return planes != nullptr && j >= 0; }
    PD size_t at(int w) const { return plane_at(SSS_WORDS, j, w); }
    PD float getf(int w) const { return __uint_as_float(planes[at(w)]); }
    PD int geti(int w) const { return (int)planes[at(w)]; }
    PD f3 get3(int w) const { return f3(getf(w), getf(w + 1), getf(w + 2)); }
    PD void putf(int w, float v) const { planes[at(w)] = __float_as_uint(v); }
    PD void puti(int w, int v) const { planes[at(w)] = (uint32_t)v; }
    PD void put3(int w, const f3 &v) const { putf(w, v.x); putf(w + 1, v.y); putf(w + 2, v.z); }
};
/* the first segment of a chain (the loop above at step 0: p = start, no error bounds, no normal, no media) */
PD void chain_begin(const ChainRef &c, int slot, int prim, const f3 &start, const f3 &pTarget) {
    const f3 d = pTarget - start;
    c.put3(SW_O, offset_ray_origin(start, f3(0.f), f3(0.f), d));
    c.put3(SW_D, d);
    c.put3(SW_TARGET, pTarget);
    c.puti(SW_SLOT, slot);
    c.puti(SW_STATE, SSS_LIVE);
    c.puti(SW_FOUND, 0);
    c.puti(SW_STEPS, 0);
    c.puti(SW_MED, si_medium(f3(0.f), -1, -1, d));
    c.puti(SW_PRIM, prim);
}
/* one iteration of the loop above for a chain whose current segment has just been traced (hp < 0: nothing hit; t: where the
 * segment ended); false: the chain has ended.  The next segment is left in the record and returned in *o_next / *d_next. */
PD bool chain_advance(const DScene &sc, const ChainRef &c, int hp, int hf, float t, f3 *o_next, f3 *d_next) {
    const f3 o = c.get3(SW_O), d = c.get3(SW_D);
    const int steps = c.geti(SW_STEPS) + 1;
    c.puti(SW_STEPS, steps);
    bool live = !(is_black(d) || hp < 0) && steps < PROBE_MAX_STEPS;
    if (live) {
        const int med = c.geti(SW_MED);
        SI si;
        make_si(sc, hp, hf, o, d, t, med, &si);
        if (hp == c.geti(SW_PRIM)) {
            const int nFound = c.geti(SW_FOUND);
            if (nFound < PROBE_KEEP) {
                const int k = SW_KEEP + nFound * SW_KEEP_WORDS;
                c.put3(k, o); c.put3(k + 3, d); c.putf(k + 6, t); c.puti(k + 7, hf); c.puti(k + 8, med);
            }
            c.puti(SW_FOUND, nFound + 1);
            if (nFound + 1 > PROBE_MAX_FOUND - 1) live = false;
        }
        if (live) {
            const f3 dn = c.get3(SW_TARGET) - si.p;
            const f3 on = offset_ray_origin(si.p, si.pError, si.n, dn);
            c.put3(SW_O, on);
            c.put3(SW_D, dn);
            c.puti(SW_MED, si_medium(si.n, si.med_in, si.med_out, dn));
            *o_next = on;
            *d_next = dn;
        }
    }
    if (!live) c.puti(SW_STATE, SSS_DONE);
    return live;
}

/* Sample_Sp up to its probe segment, bssrdf.cpp:462-505: false when the sample carries nothing (S = 0) */
PD bool bssrdf_probe_setup(const Bssrdf &b, float &u1, const f2 &u2, f3 *start, f3 *pTarget) {
    f3 vx, vy, vz;
    if (u1 < .5f) { vx = b.ss; vy = b.ts; vz = b.ns; u1 *= 2; }
    else if (u1 < .75f) { vx = b.ts; vy = b.ns; vz = b.ss; u1 = (u1 - .5f) * 4; }
    else { vx = b.ns; vy = b.ss; vz = b.ts; u1 = (u1 - .75f) * 4; }
    int ch = clampi((int)(u1 * 3), 0, 2);
    u1 = u1 * 3 - ch;
    float r = bssrdf_sample_sr(b, ch, u2.x);
    if (r < 0) return false;
    float phi = 2 * kPi * u2.y;
    float rMax = bssrdf_sample_sr(b, ch, 0.999f);
    if (r >= rMax) return false;
    float l = 2 * sqrtf(rMax * rMax - r * r);
    *start = b.po_p + r * (vx * cosf(phi) + vy * sinf(phi)) - l * vz * 0.5f;
    *pTarget = *start + l * vz;
    return true;
}

/* What the BSSRDF tail knows once the entry vertex has set up its probe: the BSSRDF, the remapped first draw that later picks
 * among the chain's hits, the probe segment (ok = false: the sample carries nothing, S = 0).  main.cpp:207-211 +
 * Sample_S / Sample_Sp up to bssrdf.cpp:505: three draws. */
struct ProbeSetup {
    Bssrdf b;
    float u1;
    f3 start, pTarget;
    bool ok;
};
PD void bssrdf_setup(const DScene &sc, const BSDF &frame, const BssrdfParams &bp, const SI &po, Rng &rng, ProbeSetup *ps) {
    bssrdf_make(sc, bp, po, frame, &ps->b);
    ps->u1 = rng.next();
    f2 u2 = rand2(rng);
    ps->start = f3(0.f);
    ps->pTarget = f3(0.f);
    ps->ok = bssrdf_probe_setup(ps->b, ps->u1, u2, &ps->start, &ps->pTarget);
}
#define BSSRDF_SETUP_DRAWS 3
/* a chain record made from a setup (k_shade<SSS_BEGIN>), and the setup read back from it (the tail kernels) */
PD void chain_put_setup(const ChainRef &c, const DScene &sc, const ProbeSetup &ps, int slot) {
    if (ps.ok) chain_begin(c, slot, ps.b.po_prim, ps.start, ps.pTarget);
    else { /* a finished chain without hits; its "segment" is never traced */
        c.puti(SW_SLOT, slot);
        c.puti(SW_STATE, SSS_DONE);
        c.puti(SW_FOUND, 0);
        c.puti(SW_STEPS, 0);
        c.puti(SW_PRIM, ps.b.po_prim);
        c.put3(SW_O, f3(0.f));
        c.put3(SW_D, f3(0.f));
        c.put3(SW_TARGET, f3(0.f));
    }
    c.puti(SW_OK, ps.ok ? 1 : 0);
    c.putf(SW_U1, ps.u1);
    c.put3(SW_START, ps.start);
    c.put3(SW_B_PO, ps.b.po_p);
    c.put3(SW_B_NS, ps.b.ns); c.put3(SW_B_SS, ps.b.ss); c.put3(SW_B_TS, ps.b.ts);
    c.putf(SW_B_ETA, ps.b.eta);
    c.put3(SW_B_SIGT, ps.b.sigma_t); c.put3(SW_B_RHO, ps.b.rho);
    c.puti(SW_B_TABLE, (int)(ps.b.t - sc.tables));
}
PD void chain_get_setup(const ChainRef &c, const DScene &sc, ProbeSetup *ps) {
    ps->ok = c.geti(SW_OK) != 0;
    ps->u1 = c.getf(SW_U1);
    ps->start = c.get3(SW_START);
    ps->pTarget = c.get3(SW_TARGET);
    ps->b.po_p = c.get3(SW_B_PO);
    ps->b.po_prim = c.geti(SW_PRIM);
    ps->b.ns = c.get3(SW_B_NS); ps->b.ss = c.get3(SW_B_SS); ps->b.ts = c.get3(SW_B_TS);
    ps->b.eta = c.getf(SW_B_ETA);
    ps->b.sigma_t = c.get3(SW_B_SIGT); ps->b.rho = c.get3(SW_B_RHO);
    ps->b.t = sc.tables + c.geti(SW_B_TABLE);
}
/* a record the tail can use without any traversal: the rounds finished the chain and the record keeps every hit */
PD bool chain_settled(const ChainRef &chain) {
    return chain.valid() && chain.geti(SW_STATE) == SSS_DONE && chain.geti(SW_FOUND) <= PROBE_KEEP;
}

/* The rest of Sample_Sp, bssrdf.cpp:506-560: the chain's outcome, the pick among its hits, pdf and Sr.
 * chain valid: the record the tracer rounds worked on.  CAN_WALK: whatever they left undone is traversed here, in the lane -
 * a chain still alive goes on from its current segment (the same chain_advance, one traversal per round), a pick beyond
 * the kept hits walks the chain again up to it - and without a record the whole chain is walked as in round 2.
 * CAN_WALK = false (k_shade<SSS_TAIL>): the caller has checked chain_settled; no traversal code is instantiated. */
template <bool CAN_WALK>
PD f3 bssrdf_exit_point(const DScene &sc, const ProbeSetup &ps, const ChainRef &chain, SI *pi, float *pdf, unsigned *probes, const LaneStack &ls) {
    *pdf = 0;
    if (!ps.ok) return f3(0.f);
    const Bssrdf &b = ps.b;
    int nFound;
    if (!CAN_WALK || chain.valid()) {
        if (CAN_WALK) {
            while (chain.geti(SW_STATE) == SSS_LIVE) {
                float tmax = 1 - kShadowEpsilon, b0, b1;
                int hp = -1, hf = 0;
                if (probes) (*probes)++;
                const bool hit = traverse<false, false>(sc, chain.get3(SW_O), chain.get3(SW_D), tmax, hp, hf, b0, b1, nullptr, ls);
                f3 on, dn;
                chain_advance(sc, chain, hit ? hp : -1, hf, tmax, &on, &dn);
            }
        }
        nFound = chain.geti(SW_FOUND);
        if (nFound == 0) return f3(0.0f);
        int selected = clampi((int)(ps.u1 * nFound), 0, nFound - 1);
        if (!CAN_WALK || selected < PROBE_KEEP) {
            const int k = SW_KEEP + selected * SW_KEEP_WORDS;
            make_si(sc, b.po_prim, chain.geti(k + 7), chain.get3(k), chain.get3(k + 3), chain.getf(k + 6), chain.geti(k + 8), pi);
        } else if (CAN_WALK) bssrdf_probe_chain(sc, ps.start, ps.pTarget, b.po_prim, selected, pi, nullptr, ls);
    } else if (CAN_WALK) {
        ProbeHit keep[PROBE_KEEP];
        nFound = bssrdf_probe_chain(sc, ps.start, ps.pTarget, b.po_prim, -1, pi, probes, ls, keep);
        if (nFound == 0) return f3(0.0f);
        int selected = clampi((int)(ps.u1 * nFound), 0, nFound - 1);
        if (selected < PROBE_KEEP) {
            ProbeHit k = keep[0];
#pragma unroll
            for (int i = 1; i < PROBE_KEEP; i++)
                if (selected == i) k = keep[i];
            make_si(sc, b.po_prim, k.face, k.o, k.d, k.t, k.med, pi);
        } else bssrdf_probe_chain(sc, ps.start, ps.pTarget, b.po_prim, selected, pi, nullptr, ls);
    } else return f3(0.f); /* (not reached: !CAN_WALK takes the first branch) */
    *pdf = bssrdf_pdf_sp(b, pi->p, pi->n) / nFound;
    return bssrdf_sr(b, length(b.po_p - pi->p));
}

/* exit: where the direct light at the exit point goes when it is not resolved in place - a second direct-light record,
 * numbered like the chain, whose two rays join the bounce's shadow / MIS launches of the persistent tracer and which
 * k_nee_resolve2 adds to L after the vertex's own term (ptr_runtime.hip).  route[j] receives the record's flags. */
struct ExitNee {
    float *planes; /* NEE_WORDS tiled planes indexed by chain number, or null: resolve in place */
    int *route;
};
/* The BSSRDF tail of Li_PathSampled / Li_VolPath (main.cpp:207-228, 320-342) from the probe's outcome on: pick the exit
 * point, account for direct light there (in place: added to *L_extra, which follows the vertex's own direct-light term in L;
 * or left as a record, see ExitNee), sample the exit direction.  bsdf: this lane's lobe column and the ENTRY point's frame
 * (`BSDF bsdf(isect)` is only emptied, main.cpp:207-228).  Returns false if the path ends; *extra_in_place tells whether
 * *L_extra was computed here. */
template <bool CAN_WALK>
PD bool bssrdf_finish(const DScene &sc, BSDF &bsdf, const ProbeSetup &ps, Rng &rng, bool handleMedium, bool walk, f3 *L_extra, bool *extra_in_place,
                      f3 &beta, int &pflags, f3 &ro, f3 &rd, int &ray_medium, NeeCtr *nctr, unsigned *probes, const LaneStack &ls, const ChainRef &chain,
                      const ExitNee &exit) {
    SI pi;
    float pdf = 0;
    bsdf.n = 0;
    *extra_in_place = false;
    f3 S = bssrdf_exit_point<CAN_WALK>(sc, ps, chain, &pi, &pdf, probes, ls);
    if (!is_black(S)) { /* Sample_S, bssrdf.cpp:447-460: the exit vertex scatters through the Sw adapter lobe */
        Lobe lb;
        lobe_clear(lb);
        lb.type = BSDF_REFLECTION | BSDF_DIFFUSE;
        lb.impl = LB_BSSRDF_ADAPTER;
        lb.C = ps.b.eta;
        lobe_push(&bsdf, lb);
        pi.wo = pi.n;
    }
    if (is_black(S) || is_zero(pdf)) return false;
    beta *= S / pdf;
    f2 u2 = rand2(rng);
    f3 u3 = rand3(rng);
    if (sc.n_lights >= 1 && (!CAN_WALK || (exit.planes && chain.valid() && !walk))) {
#ifdef SH_STATS
        const NeeToPlanes sink = {exit.planes, chain.j, false, nullptr};
#else
        const NeeToPlanes sink = {exit.planes, chain.j, false};
#endif
        const int fl = build_nee(sc, ref_from_si(pi), &bsdf, 0, u2, u3, handleMedium, beta, false, sink);
        exit.route[chain.j] = (fl ? ROUTE_NEE : 0) | ((fl & NEE_A) ? ROUTE_SHADOW : 0) | ((fl & NEE_B) ? ROUTE_MIS : 0);
    } else if (CAN_WALK && sc.n_lights >= 1) {
        NeeRec r2;
        int fl = build_nee(sc, ref_from_si(pi), &bsdf, 0, u2, u3, handleMedium, beta, false, NeeToStruct{&r2});
        *extra_in_place = true;
        if (fl) {
            if (walk) {
                if (fl & NEE_A) nee_shadow<true>(sc, &r2, *nctr, ls);
                *L_extra = nee_mis_resolve<true>(sc, r2, *nctr, ls);
            } else {
                if (fl & NEE_A) nee_shadow<false>(sc, &r2, *nctr, ls);
                *L_extra = nee_mis_resolve<false>(sc, r2, *nctr, ls);
            }
        }
    }
    u2 = rand2(rng);
    f3 wi;
    int flags = 0;
    f3 f = bsdf_sample_f(bsdf, pi.wo, &wi, u2, &pdf, BSDF_ALL, &flags);
    if (is_black(f) || is_zero(pdf)) return false;
    beta *= f * absdot(wi, pi.n) / pdf;
    if (flags & BSDF_SPECULAR) pflags |= PFL_SPECULAR;
    else pflags &= ~PFL_SPECULAR;
    ro = offset_ray_origin(pi.p, pi.pError, pi.n, wi);
    ray_medium = si_medium(pi.n, pi.med_in, pi.med_out, wi);
    rd = wi;
    return true;
}
/* the whole tail in the vertex's lane (k_finish<true>, and through it the reference's own order of events) */
PD bool bssrdf_step(const DScene &sc, BSDF &bsdf, const BssrdfParams &bp, const SI &po, Rng &rng, bool handleMedium, bool walk, f3 *L_extra,
                    f3 &beta, int &pflags, f3 &ro, f3 &rd, int &ray_medium, NeeCtr &nctr, unsigned *probes, const LaneStack &ls) {
    ProbeSetup ps;
    bssrdf_setup(sc, bsdf, bp, po, rng, &ps);
    bool in_place;
    return bssrdf_finish<true>(sc, bsdf, ps, rng, handleMedium, walk, L_extra, &in_place, beta, pflags, ro, rd, ray_medium, &nctr, probes, ls,
                               ChainRef{nullptr, -1}, ExitNee{nullptr, nullptr});
}
