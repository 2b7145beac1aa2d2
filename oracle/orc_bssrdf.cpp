/*
 * orc_bssrdf.cpp — ORACLE (test infrastructure only).  PARITY UNPINNED.
 * Restates src/core/bssrdf.cpp: Fresnel moments (:7-30), Catmull-Rom utilities (:32-191,
 * 300-357), photon-beam-diffusion table (:193-298), SubsurfaceFromDiffuse (:359-369) and the
 * tabulated SeparableBSSRDF (:383-581, 645-648, 726-735).  The Disney BSSRDF is not restated:
 * nothing calls Init_DisneyBSSRDF.
 */
#include "orc_scene.h"

namespace orc {

static Float FresnelMoment1(Float eta) {
    Float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1)
        return 0.45966f - 1.73965f * eta + 3.37668f * eta2 - 3.904945 * eta3 + 2.49277f * eta4 - 0.68441f * eta5;
    else
        return -4.61686f + 11.1136f * eta - 10.4646f * eta2 + 5.11455f * eta3 - 1.27198f * eta4 + 0.12746f * eta5;
}
static Float FresnelMoment2(Float eta) {
    Float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1) {
        return 0.27614f - 0.87350f * eta + 1.12077f * eta2 - 0.65095f * eta3 + 0.07883f * eta4 + 0.04860f * eta5;
    } else {
        Float r_eta = 1 / eta, r_eta2 = r_eta * r_eta, r_eta3 = r_eta2 * r_eta;
        return -547.033f + 45.3087f * r_eta3 - 218.725f * r_eta2 + 458.843f * r_eta + 404.557f * eta - 189.519f * eta2 +
               54.9327f * eta3 - 9.00603f * eta4 + 0.63942f * eta5;
    }
}

static bool CatmullRomWeights(int size, const Float *nodes, Float x, int *offset, Float *weights) {
    if (!(x >= nodes[0] && x <= nodes[size - 1])) return false;
    int idx = FindInterval(size, [&](int i) { return nodes[i] <= x; });
    *offset = idx - 1;
    Float x0 = nodes[idx], x1 = nodes[idx + 1];
    Float t = (x - x0) / (x1 - x0), t2 = t * t, t3 = t2 * t;
    weights[1] = 2 * t3 - 3 * t2 + 1;
    weights[2] = -2 * t3 + 3 * t2;
    if (idx > 0) {
        Float w0 = (t3 - 2 * t2 + t) * (x1 - x0) / (x1 - nodes[idx - 1]);
        weights[0] = -w0;
        weights[2] += w0;
    } else {
        Float w0 = t3 - 2 * t2 + t;
        weights[0] = 0;
        weights[1] -= w0;
        weights[2] += w0;
    }
    if (idx + 2 < size) {
        Float w3 = (t3 - t2) * (x1 - x0) / (nodes[idx + 2] - x0);
        weights[1] -= w3;
        weights[3] = w3;
    } else {
        Float w3 = t3 - t2;
        weights[1] -= w3;
        weights[2] += w3;
        weights[3] = 0;
    }
    return true;
}

static Float IntegrateCatmullRom(int n, const Float *x, const Float *values, Float *cdf) {
    Float sum = 0;
    cdf[0] = 0;
    for (int i = 0; i < n - 1; ++i) {
        Float x0 = x[i], x1 = x[i + 1];
        Float f0 = values[i], f1 = values[i + 1];
        Float width = x1 - x0;
        Float d0, d1;
        if (i > 0) d0 = width * (f1 - values[i - 1]) / (x1 - x[i - 1]);
        else d0 = f1 - f0;
        if (i + 2 < n) d1 = width * (values[i + 2] - f0) / (x[i + 2] - x0);
        else d1 = f1 - f0;
        sum += ((d0 - d1) * (1.f / 12.f) + (f0 + f1) * .5f) * width;
        cdf[i + 1] = sum;
    }
    return sum;
}

static Float SampleCatmullRom2D(int size1, int size2, const Float *nodes1, const Float *nodes2, const Float *values,
                                const Float *cdf, Float alpha, Float u) {
    int offset;
    Float weights[4];
    if (!CatmullRomWeights(size1, nodes1, alpha, &offset, weights)) return 0;
    auto interpolate = [&](const Float *array, int idx) {
        Float value = 0;
        for (int i = 0; i < 4; ++i)
            if (weights[i] != 0) value += array[(offset + i) * size2 + idx] * weights[i];
        return value;
    };
    Float maximum = interpolate(cdf, size2 - 1);
    u *= maximum;
    int idx = FindInterval(size2, [&](int i) { return interpolate(cdf, i) <= u; });
    Float f0 = interpolate(values, idx), f1 = interpolate(values, idx + 1);
    Float x0 = nodes2[idx], x1 = nodes2[idx + 1];
    Float width = x1 - x0;
    Float d0, d1;
    u = (u - interpolate(cdf, idx)) / width;
    if (idx > 0) d0 = width * (f1 - interpolate(values, idx - 1)) / (x1 - nodes2[idx - 1]);
    else d0 = f1 - f0;
    if (idx + 2 < size2) d1 = width * (interpolate(values, idx + 2) - f0) / (nodes2[idx + 2] - x0);
    else d1 = f1 - f0;
    Float t;
    if (f0 != f1) t = (f0 - sqrtf(Max((Float)0, f0 * f0 + 2 * u * (f1 - f0)))) / (f0 - f1);
    else t = u / f0;
    Float a = 0, b = 1, Fhat, fhat;
    while (true) {
        if (!(t >= a && t <= b)) t = 0.5f * (a + b);
        Fhat = t * (f0 + t * (.5f * d0 + t * ((1.f / 3.f) * (-2 * d0 - d1) + f1 - f0 + t * (.25f * (d0 + d1) + .5f * (f0 - f1)))));
        fhat = f0 + t * (d0 + t * (-2 * d0 - d1 + 3 * (f1 - f0) + t * (d0 + d1 + 2 * (f0 - f1))));
        if (Absf(Fhat - u) < 1e-6f || b - a < 1e-6f) break;
        if (Fhat - u < 0) a = t;
        else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + width * t;
}

static Float PhaseHG_(Float cosTheta, Float g) { /* medium.cpp:4-19 */
    Float denom = 1 + g * g + 2 * g * cosTheta;
    return ORC_Inv4Pi * (1 - g * g) / (denom * sqrtf(denom));
}

static Float BeamDiffusionMS(Float sigma_s, Float sigma_a, Float g, Float eta, Float r) {
    const int nSamples = 100;
    Float Ed = 0;
    Float sigmap_s = sigma_s * (1 - g);
    Float sigmap_t = sigma_a + sigmap_s;
    Float rhop = sigmap_s / sigmap_t;
    Float D_g = (2 * sigma_a + sigmap_s) / (3 * sigmap_t * sigmap_t);
    Float sigma_tr = sqrtf(sigma_a / D_g);
    Float fm1 = FresnelMoment1(eta), fm2 = FresnelMoment2(eta);
    Float ze = -2 * D_g * (1 + 3 * fm2) / (1 - 2 * fm1);
    Float cPhi = .25f * (1 - 2 * fm1), cE = .5f * (1 - 3 * fm2);
    for (int i = 0; i < nSamples; ++i) {
        Float zr = -logf(1 - (i + .5f) / nSamples) / sigmap_t;
        Float zv = -zr + 2 * ze;
        Float dr = sqrtf(r * r + zr * zr), dv = sqrtf(r * r + zv * zv);
        Float phiD = ORC_Inv4Pi / D_g * (expf(-sigma_tr * dr) / dr - expf(-sigma_tr * dv) / dv);
        Float EDn = ORC_Inv4Pi * (zr * (1 + sigma_tr * dr) * expf(-sigma_tr * dr) / (dr * dr * dr) -
                                  zv * (1 + sigma_tr * dv) * expf(-sigma_tr * dv) / (dv * dv * dv));
        Float E = phiD * cPhi + EDn * cE;
        Float kappa = 1 - expf(-2 * sigmap_t * (dr + zr));
        Ed += kappa * rhop * rhop * E;
    }
    return Ed / nSamples;
}

static Float BeamDiffusionSS(Float sigma_s, Float sigma_a, Float g, Float eta, Float r) {
    Float sigma_t = sigma_a + sigma_s, rho = sigma_s / sigma_t;
    Float tCrit = r * sqrtf(eta * eta - 1);
    Float Ess = 0;
    const int nSamples = 100;
    for (int i = 0; i < nSamples; ++i) {
        Float ti = tCrit - logf(1 - (i + .5f) / nSamples) / sigma_t;
        Float d = sqrtf(r * r + ti * ti);
        Float cosThetaO = ti / d;
        Ess += rho * expf(-sigma_t * (d + tCrit)) / (d * d) * PhaseHG_(cosThetaO, g) * (1 - FrDieletric(-cosThetaO, 1, eta)) *
               std::abs(cosThetaO);
    }
    return Ess / nSamples;
}

void ComputeBeamDiffusionBSSRDF(Float g, Float eta, BSSRDFTable *t) {
    t->rhoSamples.assign(t->nRhoSamples, 0);
    t->radiusSamples.assign(t->nRadiusSamples, 0);
    t->profile.assign((size_t)t->nRhoSamples * t->nRadiusSamples, 0);
    t->rhoEff.assign(t->nRhoSamples, 0);
    t->profileCDF.assign((size_t)t->nRhoSamples * t->nRadiusSamples, 0);
    t->radiusSamples[0] = 0;
    t->radiusSamples[1] = 2.5e-3f;
    for (int i = 2; i < t->nRadiusSamples; ++i) t->radiusSamples[i] = t->radiusSamples[i - 1] * 1.2f;
    for (int i = 0; i < t->nRhoSamples; ++i)
        t->rhoSamples[i] = (1 - expf(-8 * i / (Float)(t->nRhoSamples - 1))) / (1 - expf(-8.f));
    for (int i = 0; i < t->nRhoSamples; i++) {
        for (int j = 0; j < t->nRadiusSamples; ++j) {
            Float rho = t->rhoSamples[i], r = t->radiusSamples[j];
            t->profile[i * t->nRadiusSamples + j] =
                2 * ORC_Pi * r * (BeamDiffusionSS(rho, 1 - rho, g, eta, r) + BeamDiffusionMS(rho, 1 - rho, g, eta, r));
        }
        t->rhoEff[i] = IntegrateCatmullRom(t->nRadiusSamples, t->radiusSamples.data(), &t->profile[i * t->nRadiusSamples],
                                           &t->profileCDF[i * t->nRadiusSamples]);
    }
}

static Float InvertCatmullRom(int n, const Float *x, const Float *values, Float u) {
    if (!(u > values[0])) return x[0];
    else if (!(u < values[n - 1])) return x[n - 1];
    int i = FindInterval(n, [&](int i) { return values[i] <= u; });
    Float x0 = x[i], x1 = x[i + 1];
    Float f0 = values[i], f1 = values[i + 1];
    Float width = x1 - x0;
    Float d0, d1;
    if (i > 0) d0 = width * (f1 - values[i - 1]) / (x1 - x[i - 1]);
    else d0 = f1 - f0;
    if (i + 2 < n) d1 = width * (values[i + 2] - f0) / (x[i + 2] - x0);
    else d1 = f1 - f0;
    Float a = 0, b = 1, t = .5f;
    Float Fhat, fhat;
    while (true) {
        if (!(t > a && t < b)) t = 0.5f * (a + b);
        Float t2 = t * t, t3 = t2 * t;
        Fhat = (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
        fhat = (6 * t2 - 6 * t) * f0 + (-6 * t2 + 6 * t) * f1 + (3 * t2 - 4 * t + 1) * d0 + (3 * t2 - 2 * t) * d1;
        if (std::abs(Fhat - u) < 1e-6f || b - a < 1e-6f) break;
        if (Fhat - u < 0) a = t;
        else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + t * width;
}

void SubsurfaceFromDiffuse(const BSSRDFTable &t, const V3 &rhoEff, const V3 &mfp, V3 *sigma_a, V3 *sigma_s) {
    for (int c = 0; c < 3; ++c) {
        Float rho = InvertCatmullRom(t.nRhoSamples, t.rhoSamples.data(), t.rhoEff.data(), rhoEff[c]);
        (*sigma_s)[c] = rho / mfp[c];
        (*sigma_a)[c] = (1 - rho) / mfp[c];
    }
}

/* bssrdf.h:108-117 + bssrdf.cpp:726-735 */
void InitTabulatedBSSRDF(SeparableBSSRDF *b, const SurfaceInteraction &po, Float eta, int material, const V3 &sigma_a,
                         const V3 &sigma_s, const BSSRDFTable *table) {
    b->po = po;
    b->eta = eta;
    b->ns = po.n;
    b->ss = Normalize(po.dpdu);
    b->ts = Cross(b->ns, b->ss);
    b->material = material;
    b->table = table;
    b->sigma_t = sigma_a + sigma_s;
    for (int c = 0; c < 3; ++c) b->rho[c] = !IsZero(b->sigma_t[c]) ? (sigma_s[c] / b->sigma_t[c]) : 0;
}

/* bssrdf.h:126-129 */
V3 BSSRDF_Sw(const SeparableBSSRDF &b, const V3 &w) {
    Float c = 1 - 2 * FresnelMoment1(1 / b.eta);
    return V3((Float)((1 - FrDieletric(CosTheta(w), 1, b.eta)) / (c * ORC_Pi)));
}

static V3 Tabulated_Sr(const SeparableBSSRDF &b, Float r) { /* :383-411 */
    const BSSRDFTable *table = b.table;
    V3 Sr(0.f);
    for (int ch = 0; ch < 3; ++ch) {
        Float rOptical = r * b.sigma_t[ch];
        int rhoOffset, radiusOffset;
        Float rhoWeights[4], radiusWeights[4];
        if (!CatmullRomWeights(table->nRhoSamples, table->rhoSamples.data(), b.rho[ch], &rhoOffset, rhoWeights) ||
            !CatmullRomWeights(table->nRadiusSamples, table->radiusSamples.data(), rOptical, &radiusOffset, radiusWeights))
            continue;
        Float sr = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                Float weight = rhoWeights[i] * radiusWeights[j];
                if (!IsZero(weight)) sr += weight * table->EvalProfile(rhoOffset + i, radiusOffset + j);
            }
        if (!IsZero(rOptical)) sr /= 2 * ORC_Pi * rOptical;
        Sr[ch] = sr;
    }
    Sr *= b.sigma_t * b.sigma_t;
    return V3(Clampf(Sr.x, -1, 1), Clampf(Sr.y, -1, 1), Clampf(Sr.z, -1, 1)); /* Clamp(Sr): [-1,1] */
}
static Float Tabulated_Pdf_Sr(const SeparableBSSRDF &b, int ch, Float r) { /* :413-437 */
    const BSSRDFTable *table = b.table;
    Float rOptical = r * b.sigma_t[ch];
    int rhoOffset, radiusOffset;
    Float rhoWeights[4], radiusWeights[4];
    if (!CatmullRomWeights(table->nRhoSamples, table->rhoSamples.data(), b.rho[ch], &rhoOffset, rhoWeights) ||
        !CatmullRomWeights(table->nRadiusSamples, table->radiusSamples.data(), rOptical, &radiusOffset, radiusWeights))
        return 0.f;
    Float sr = 0, rhoEff = 0;
    for (int i = 0; i < 4; ++i) {
        if (IsZero(rhoWeights[i])) continue;
        rhoEff += table->rhoEff[rhoOffset + i] * rhoWeights[i];
        for (int j = 0; j < 4; ++j) {
            if (IsZero(radiusWeights[j])) continue;
            sr += table->EvalProfile(rhoOffset + i, radiusOffset + j) * rhoWeights[i] * radiusWeights[j];
        }
    }
    if (!IsZero(rOptical)) sr /= 2 * ORC_Pi * rOptical;
    return Max((Float)0, sr * b.sigma_t[ch] * b.sigma_t[ch] / rhoEff);
}
static Float Tabulated_Sample_Sr(const SeparableBSSRDF &b, int ch, Float u) { /* :439-445 */
    if (IsZero(b.sigma_t[ch])) return -1;
    const BSSRDFTable *t = b.table;
    return SampleCatmullRom2D(t->nRhoSamples, t->nRadiusSamples, t->rhoSamples.data(), t->radiusSamples.data(), t->profile.data(),
                              t->profileCDF.data(), b.rho[ch], u) /
           b.sigma_t[ch];
}
static Float Pdf_Sp(const SeparableBSSRDF &b, const SurfaceInteraction *pi) { /* :562-581 */
    V3 d = b.po.p - pi->p;
    V3 dLocal(Dot(b.ss, d), Dot(b.ts, d), Dot(b.ns, d));
    V3 nLocal(Dot(b.ss, pi->n), Dot(b.ts, pi->n), Dot(b.ns, pi->n));
    Float rProj[3] = {sqrtf(dLocal.y * dLocal.y + dLocal.z * dLocal.z), sqrtf(dLocal.z * dLocal.z + dLocal.x * dLocal.x),
                      sqrtf(dLocal.x * dLocal.x + dLocal.y * dLocal.y)};
    Float pdf = 0, axisProb[3] = {.25f, .25f, .5f};
    Float chProb = 1 / (Float)3;
    for (int axis = 0; axis < 3; ++axis)
        for (int ch = 0; ch < 3; ++ch) pdf += Tabulated_Pdf_Sr(b, ch, rProj[axis]) * Absf(nLocal[axis]) * chProb * axisProb[axis];
    return pdf;
}

/* interaction.cpp:12-16: the Point3f overload keeps d = p2 - p (not target - origin) */
static Ray SpawnRayToPoint(const Interaction &it, const V3 &p2) {
    V3 origin = OffsetRayOrigin(it.p, it.pError, it.n, p2 - it.p);
    V3 d = p2 - it.p;
    return Ray(origin, d, 1 - ORC_ShadowEpsilon, 0.f, it.GetMedium(d));
}

/* :462-560 */
static V3 Sample_Sp(const Scene &sc, const SeparableBSSRDF &b, Float u1, const V2 &u2, SurfaceInteraction *pi, Float *pdf) {
    V3 vx, vy, vz;
    *pdf = 0;
    if (u1 < .5f) { vx = b.ss; vy = b.ts; vz = b.ns; u1 *= 2; }
    else if (u1 < .75f) { vx = b.ts; vy = b.ns; vz = b.ss; u1 = (u1 - .5f) * 4; }
    else { vx = b.ns; vy = b.ss; vz = b.ts; u1 = (u1 - .75f) * 4; }
    int ch = Clampi((int)(u1 * 3), 0, 2);
    u1 = u1 * 3 - ch;
    Float r = Tabulated_Sample_Sr(b, ch, u2[0]);
    if (r < 0) return V3(0.f);
    Float phi = 2 * ORC_Pi * u2[1];
    Float rMax = Tabulated_Sample_Sr(b, ch, 0.999f);
    if (r >= rMax) return V3(0.f);
    Float l = 2 * sqrtf(rMax * rMax - r * r);
    Interaction base;
    base.p = b.po.p + r * (vx * cosf(phi) + vy * sinf(phi)) - l * vz * 0.5f;
    V3 pTarget = base.p + l * vz;
    const int maxSi = 256;
    std::vector<SurfaceInteraction> si(maxSi);
    int nFound = 0;
    SurfaceInteraction *ptr = &si[nFound];
    /* the reference's loop is unbounded; oracle and product both stop a chain stuck on a degenerate hit after 4096 segments */
    for (int step = 0; step < 4096; step++) {
        Ray ray = SpawnRayToPoint(base, pTarget);
        g_ctr.probe_rays++;
        bool hit = SceneIntersect(sc, ray, ptr);
        if (IsBlack(ray.d) || !hit) break;
        base = *ptr;
        if (ptr->primitive == b.po.primitive) {
            nFound++;
            if (nFound > maxSi - 1) break;
            ptr = &si[nFound];
        }
    }
    if (nFound == 0) return V3(0.0f);
    int selected = Clampi((int)(u1 * nFound), 0, nFound - 1);
    *pi = si[selected];
    *pdf = Pdf_Sp(b, pi) / nFound;
    return Tabulated_Sr(b, Distance(b.po.p, pi->p));
}

/* :447-460.  The adapter lobe keeps a pointer to bsdf->bssrdf, as Init_BSSRDF(this) does. */
V3 BSSRDF_Sample_S(const Scene &sc, const SeparableBSSRDF &b, BSDF *bsdf, Float u1, const V2 &u2, SurfaceInteraction *si,
                   Float *pdf) {
    V3 Sp = Sample_Sp(sc, b, u1, u2, si, pdf);
    if (!IsBlack(Sp)) {
        BxDF bx;
        bx.impl = BSSRDFAdapter;
        bx.type = BSDF_REFLECTION | BSDF_DIFFUSE;
        bx.bssrdf = &b;
        bsdf->Push(bx);
        si->wo = si->n;
    }
    return Sp;
}

/* ---- unit hooks (tests only) ---- */
V3 Unit_Sr(const SeparableBSSRDF &b, Float r) { return Tabulated_Sr(b, r); }
Float Unit_Pdf_Sr(const SeparableBSSRDF &b, int ch, Float r) { return Tabulated_Pdf_Sr(b, ch, r); }
Float Unit_Sample_Sr(const SeparableBSSRDF &b, int ch, Float u) { return Tabulated_Sample_Sr(b, ch, u); }

} // namespace orc
