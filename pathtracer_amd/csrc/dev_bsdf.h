/*
 * dev_bsdf.h — BSDF construction, evaluation and sampling on gfx950.
 * Reference: src/core/reflection.cpp:36-701 (lobes, BSDF), src/core/microfacet.cpp:3-120,
 * src/core/material.cpp:5-86 (Fresnel), src/materials/ (all files) (material -> lobes),
 * src/core/texture.h:18-84 (textures).  The reference's BSDF holds 8 x 144-byte tagged-union
 * BxDFs by value (1480 B); here a lobe is 80 B and a BSDF holds at most 5 (the most any
 * material pushes: Uber), which keeps the whole BSDF in registers / a small scratch frame.
 */
#pragma once
#include "dev_isect.h"

enum { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };
enum { LB_LAMBERT_R = 0, LB_LAMBERT_T, LB_SPEC_R, LB_SPEC_T, LB_FRESNEL_SPEC, LB_OREN_NAYAR, LB_MICRO_T, LB_MICRO_R, LB_FRESNEL_BLEND,
       LB_BSSRDF_ADAPTER, LB_DISNEY_DIFFUSE, LB_DISNEY_FAKESS, LB_DISNEY_RETRO, LB_DISNEY_SHEEN, LB_DISNEY_CLEARCOAT, LB_INVALID };
enum { FR_NOOP = 0, FR_DIELECTRIC = 1, FR_CONDUCTOR = 2 };
enum { MT_MATTE = 1, MT_MIRROR, MT_GLASS, MT_METAL, MT_TRANSLUCENT, MT_PLASTIC, MT_UBER, MT_SUBSURFACE, MT_KDSUBSURFACE, MT_PLAYGROUND, MT_DISNEY };
#ifndef MAX_LOBES
#define MAX_LOBES 5
#endif
/* The three per-lobe switches are called from a dozen places of a shade kernel; kept out of line (one
 * copy each): fewer spills around them (114 instead of 194 VGPRs), half the compile time, shade 6 % faster. */
#define LOBE_FN static __device__ __noinline__

struct Lobe { /* 16 words; a field is written only by the lobe kinds that read it */
    int type, impl;
    f3 S;         /* reflectance (BxDF::S) */
    f3 T;         /* transmittance (BxDF::T); conductor lobes: etaT spectrum (etaI is always 1) */
    float A, B;   /* eta pair, or Oren-Nayar A,B; conductor lobes: k.x, k.y; Disney: roughness (fake-SS, retro), clearcoat weight / gloss alpha */
    float C;      /* conductor lobes: k.z; BSSRDF adapter: eta of the owning BSSRDF; Disney sheen: sheenTint (T = sheen) */
    float ax, ay; /* Trowbridge-Reitz alphas */
    int fr_type;
    float fr_etaI, fr_etaT;
};
/* The lobes of a BSDF live in LDS, not in the struct: a lobe array indexed at run time would sit in scratch
 * memory, and the ~300 lobe words a vertex reads back (every lobe, for f, pdf and the sample) then come from
 * L2 / Infinity Cache (8 waves x 64 lanes x 320 B per CU is far beyond the L1).  The kernel lends each lane a
 * column of a __shared__ array: word w of lobe i is lm[(i * 16 + w) * LOBE_STRIDE] (LOBE_STRIDE = threads per
 * block, so a wavefront's accesses are conflict-free).  Lobes are read into registers to be evaluated. */
#define LOBE_WORDS 16
#define LOBE_STRIDE 256 /* = TPB of the kernels that shade (rt_common.h) */
#define LOBE_LDS_WORDS (MAX_LOBES * LOBE_WORDS * LOBE_STRIDE)
struct BSDF {
    float *lm; /* this lane's lobe column (set by the kernel before bsdf_init) */
    int n;
    f3 ns, ss, ts; /* ng == ns (reflection.h:367-372) */
};
PD int lobe_type(const BSDF &b, int i) { return __float_as_int(b.lm[(i * LOBE_WORDS) * LOBE_STRIDE]); }
PD Lobe lobe_load(const BSDF &b, int i) {
    const float *p = b.lm + (i * LOBE_WORDS) * LOBE_STRIDE;
    Lobe l;
    l.type = __float_as_int(p[0]); l.impl = __float_as_int(p[1 * LOBE_STRIDE]);
    l.S = f3(p[2 * LOBE_STRIDE], p[3 * LOBE_STRIDE], p[4 * LOBE_STRIDE]);
    l.T = f3(p[5 * LOBE_STRIDE], p[6 * LOBE_STRIDE], p[7 * LOBE_STRIDE]);
    l.A = p[8 * LOBE_STRIDE]; l.B = p[9 * LOBE_STRIDE]; l.C = p[10 * LOBE_STRIDE];
    l.ax = p[11 * LOBE_STRIDE]; l.ay = p[12 * LOBE_STRIDE];
    l.fr_type = __float_as_int(p[13 * LOBE_STRIDE]); l.fr_etaI = p[14 * LOBE_STRIDE]; l.fr_etaT = p[15 * LOBE_STRIDE];
    return l;
}
PD void lobe_push(BSDF *b, const Lobe &l) {
    float *p = b->lm + (b->n * LOBE_WORDS) * LOBE_STRIDE;
    b->n += 1;
    p[0] = __int_as_float(l.type); p[1 * LOBE_STRIDE] = __int_as_float(l.impl);
    p[2 * LOBE_STRIDE] = l.S.x; p[3 * LOBE_STRIDE] = l.S.y; p[4 * LOBE_STRIDE] = l.S.z;
    p[5 * LOBE_STRIDE] = l.T.x; p[6 * LOBE_STRIDE] = l.T.y; p[7 * LOBE_STRIDE] = l.T.z;
    p[8 * LOBE_STRIDE] = l.A; p[9 * LOBE_STRIDE] = l.B; p[10 * LOBE_STRIDE] = l.C;
    p[11 * LOBE_STRIDE] = l.ax; p[12 * LOBE_STRIDE] = l.ay;
    p[13 * LOBE_STRIDE] = __int_as_float(l.fr_type); p[14 * LOBE_STRIDE] = l.fr_etaI; p[15 * LOBE_STRIDE] = l.fr_etaT;
}

PD bool type_matches(int type, int t) { return (type & t) == type; }

/* ---- Fresnel, material.cpp:5-73 ---- */
PD float fr_dielectric(float cosThetaI, float etaI, float etaT) {
    cosThetaI = clampr(cosThetaI, -1, 1);
    bool entering = cosThetaI > 0.f;
    if (!entering) {
        float s = etaI; etaI = etaT; etaT = s;
        cosThetaI = absr(cosThetaI);
    }
    float sinThetaI = sqrtf(fmaxr(0.f, 1 - cosThetaI * cosThetaI));
    float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    float cosThetaT = sqrtf(fmaxr(0.f, 1 - sinThetaT * sinThetaT));
    float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}
PD f3 fr_conductor(float cosThetaI, const f3 &etai, const f3 &etat, const f3 &k) {
    cosThetaI = clampr(cosThetaI, -1, 1);
    f3 eta = divv(etat, etai);
    f3 etak = divv(k, etai);
    float cosThetaI2 = cosThetaI * cosThetaI;
    float sinThetaI2 = 1. - cosThetaI2;
    f3 eta2 = eta * eta;
    f3 etak2 = etak * etak;
    f3 t0 = eta2 - etak2 - f3(sinThetaI2);
    f3 a2plusb2 = sqrt3(t0 * t0 + 4.0f * eta2 * etak2);
    f3 t1 = a2plusb2 + f3(cosThetaI2);
    f3 a = sqrt3(0.5f * (a2plusb2 + t0));
    f3 t2 = 2.0f * cosThetaI * a;
    f3 Rs = divv(t1 - t2, t1 + t2);
    f3 t3 = cosThetaI2 * a2plusb2 + f3(sinThetaI2 * sinThetaI2);
    f3 t4 = t2 * sinThetaI2;
    f3 Rp = divv(Rs * (t3 - t4), t3 + t4);
    return 0.5f * (Rp + Rs);
}
PD f3 fresnel_eval(const Lobe &b, float cosI) {
    if (b.fr_type == FR_DIELECTRIC) return f3(fr_dielectric(cosI, b.fr_etaI, b.fr_etaT));
    if (b.fr_type == FR_CONDUCTOR) return fr_conductor(absr(cosI), f3(1.f), b.T, f3(b.A, b.B, b.C));
    return f3(1.f);
}

/* ---- Trowbridge-Reitz, microfacet.h:30-43, microfacet.cpp ---- */
PD float roughness_to_alpha(float roughness) {
    roughness = fmaxr(roughness, (float)1e-3);
    float x = logf(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
PD float mf_lambda(float ax, float ay, const f3 &w) {
    float absTanTheta = absr(tan_theta(w));
    if (isinf(absTanTheta)) return 0.;
    float alpha = sqrtf(cos2_phi(w) * ax * ax + sin2_phi(w) * ay * ay);
    float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
    return (-1 + sqrtf(1.f + alpha2Tan2Theta)) / 2;
}
PD float mf_D(float ax, float ay, const f3 &wh) {
    float tan2Theta = tan2_theta(wh);
    if (isinf(tan2Theta)) return 0.;
    const float cos4Theta = cos2_theta(wh) * cos2_theta(wh);
    float e = (cos2_phi(wh) / (ax * ax) + sin2_phi(wh) / (ay * ay)) * tan2Theta;
    return 1 / (kPi * ax * ay * cos4Theta * (1 + e) * (1 + e));
}
PD float mf_G1(float ax, float ay, const f3 &w) { return 1 / (1 + mf_lambda(ax, ay, w)); }
PD float mf_G(float ax, float ay, const f3 &wo, const f3 &wi) { return 1 / (1 + mf_lambda(ax, ay, wo) + mf_lambda(ax, ay, wi)); }
PD float mf_pdf(float ax, float ay, const f3 &wo, const f3 &wh) { /* sampleVisibleArea is always true */
    return mf_D(ax, ay, wh) * mf_G1(ax, ay, wo) * absdot(wo, wh) / abs_cos_theta(wo);
}
PD void mf_sample11(float cosTheta, float U1, float U2, float *slope_x, float *slope_y) {
    if (cosTheta > .9999) {
        float r = sqrtf(U1 / (1 - U1));
        float phi = 6.28318530718 * U2;
        *slope_x = r * cosf(phi);
        *slope_y = r * sinf(phi);
        return;
    }
    float sinTheta = sqrtf(fmaxr(0.f, 1.f - cosTheta * cosTheta));
    float tanTheta = sinTheta / cosTheta;
    float a = 1 / tanTheta;
    float G1 = 2 / (1 + sqrtf(1.f + 1.f / (a * a)));
    float A = 2 * U1 / G1 - 1;
    float tmp = 1.f / (A * A - 1.f);
    if (tmp > 1e10) tmp = 1e10;
    float B = tanTheta;
    float D = sqrtf(fmaxr(float(B * B * tmp * tmp - (A * A - B * B) * tmp), 0.f));
    float slope_x_1 = B * tmp - D;
    float slope_x_2 = B * tmp + D;
    *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
    float S;
    if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
    else { S = -1.f; U2 = 2.f * (.5f - U2); }
    float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) / (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    *slope_y = S * z * sqrtf(1.f + *slope_x * *slope_x);
}
PD f3 mf_sample_wh(float ax, float ay, const f3 &wo, const f2 &u) {
    bool flip = wo.z < 0;
    f3 wi = flip ? -wo : wo;
    f3 wiS = normalize(f3(ax * wi.x, ay * wi.y, wi.z));
    float slope_x, slope_y;
    mf_sample11(cos_theta(wiS), u.x, u.y, &slope_x, &slope_y);
    float tmp = cos_phi(wiS) * slope_x - sin_phi(wiS) * slope_y;
    slope_y = sin_phi(wiS) * slope_x + cos_phi(wiS) * slope_y;
    slope_x = tmp;
    slope_x = ax * slope_x;
    slope_y = ay * slope_y;
    f3 wh = normalize(f3(-slope_x, -slope_y, 1.));
    if (flip) wh = -wh;
    return wh;
}

/* ---- lobes, reflection.cpp:36-589 ---- */
PD float pow5(float v) { return v * v * v * v * v; }
/* Sw of the separable BSSRDF, bssrdf.h:126-129 (FresnelMoment1: bssrdf.cpp:7-16) */
PD float fresnel_moment1(float eta) {
    float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1) return 0.45966f - 1.73965f * eta + 3.37668f * eta2 - 3.904945 * eta3 + 2.49277f * eta4 - 0.68441f * eta5;
    return -4.61686f + 11.1136f * eta - 10.4646f * eta2 + 5.11455f * eta3 - 1.27198f * eta4 + 0.12746f * eta5;
}
PD f3 bssrdf_sw(float eta, const f3 &w) {
    float c = 1 - 2 * fresnel_moment1(1 / eta);
    return f3((float)((1 - fr_dielectric(cos_theta(w), 1, eta)) / (c * kPi)));
}

/* ---- Disney lobes, reflection.cpp:3-26, 77-84, 232-295, 504-521 ---- */
PD float schlick_weight(float cosTheta) {
    float m = clampr(1 - cosTheta, 0, 1);
    return (m * m) * (m * m) * m;
}
PD float gtr1(float cosTheta, float alpha) { /* Pi is a double literal in the reference: the quotient is taken in double */
    if (alpha > 1) return (float)kInvPi;
    float alpha2 = alpha * alpha;
    return (float)((alpha2 - 1) / (kPi * logf(alpha2) * (1 + (alpha2 - 1) * cosTheta * cosTheta)));
}
PD float smith_g_ggx(float cosTheta, float alpha) {
    float alpha2 = alpha * alpha;
    float cosTheta2 = cosTheta * cosTheta;
    return 2.f / (1.f + sqrtf(alpha2 + (1.0f - alpha2) * cosTheta2));
}
PD float disney_clearcoat_pdf(const Lobe &b, const f3 &wo, const f3 &wi) {
    if (!same_hemisphere(wo, wi)) return 0;
    f3 wh = wi + wo;
    if (is_black(wh)) return 0;
    wh = normalize(wh);
    float Dr = gtr1(abs_cos_theta(wh), b.B);
    return Dr * abs_cos_theta(wh) / (4 * dot(wo, wh));
}

/* ---- MicrofacetReflection / MicrofacetTransmission (reflection.cpp) in ONE body each for Pdf and f ----
 * A rough dielectric (glass material) carries both lobes and a wavefront's lanes pick between them at random, so the two
 * `case`s ran one after the other for half the lanes each - normalize, D, G1 / G, Fresnel twice.  Here the two lobes share
 * those evaluations and differ by selects on their inputs and by their (short) final expressions; every lane still performs
 * exactly the operations of its own lobe, in the reference's order: the same floats (films bit-identical). */
PD float micro_pdf(const Lobe &b, const f3 &wo, const f3 &wi, bool refl, const f3 *wh_sampled = nullptr) {
    if (!wh_sampled && (refl ? !same_hemisphere(wo, wi) : same_hemisphere(wo, wi))) return 0;
    const float eta = cos_theta(wo) > 0 ? (b.B / b.A) : (b.A / b.B); /* (transmission only) */
    f3 wh;
    if (wh_sampled) wh = *wh_sampled;
    else {
        const f3 wis = refl ? wi : wi * eta;
        wh = normalize(wo + wis);
    }
    const float mf = mf_pdf(b.ax, b.ay, wo, wh);
    if (refl) {
        const float dww = dot(wo, wh);
        return mf / (4 * dww);
    }
    const float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
    const float dwh_dwi = absr((eta * eta * dot(wi, wh)) / (sqrtDenom * sqrtDenom));
    return mf * dwh_dwi;
}
PD f3 micro_f(const Lobe &b, const f3 &wo, const f3 &wi, bool refl) {
    if (!refl && same_hemisphere(wo, wi)) return f3(0.f);
    const float cosThetaO = refl ? abs_cos_theta(wo) : cos_theta(wo), cosThetaI = refl ? abs_cos_theta(wi) : cos_theta(wi);
    if (is_zero(cosThetaI) || is_zero(cosThetaO)) return f3(0.f);
    const float eta = cos_theta(wo) > 0 ? (b.B / b.A) : (b.A / b.B); /* (transmission only) */
    const f3 wis = refl ? wi : wi * eta;
    f3 wh = wo + wis; /* reflection: wi + wo, the same sum */
    if (refl && is_zero(wh.x) && is_zero(wh.y) && is_zero(wh.z)) return f3(0.f);
    wh = normalize(wh);
    if (!refl && wh.z < 0) wh = -wh;
    const float cosF = refl ? dot(wi, faceforward(wh, f3(0, 0, 1))) : dot(wo, wh);
    const f3 F = fresnel_eval(b, cosF);
    const float D = mf_D(b.ax, b.ay, wh), G = mf_G(b.ax, b.ay, wo, wi);
    if (refl) return b.S * D * G * F / (4 * cosThetaI * cosThetaO);
    const float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
    const float factor = 1 / eta;
    return (f3(1.f) - F) * b.T * absr(D * G * eta * eta * absdot(wi, wh) * absdot(wo, wh) * factor * factor / (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
}

PD float lobe_pdf_body(const Lobe &b, const f3 &wo, const f3 &wi) {
    switch (b.impl) {
    case LB_LAMBERT_R: case LB_OREN_NAYAR: case LB_BSSRDF_ADAPTER:
    case LB_DISNEY_DIFFUSE: case LB_DISNEY_FAKESS: case LB_DISNEY_RETRO: case LB_DISNEY_SHEEN: return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * kInvPi : 0;
    case LB_DISNEY_CLEARCOAT: return disney_clearcoat_pdf(b, wo, wi);
    case LB_LAMBERT_T: return !same_hemisphere(wo, wi) ? abs_cos_theta(wi) * kInvPi : 0;
    case LB_MICRO_T: case LB_MICRO_R: return micro_pdf(b, wo, wi, b.impl == LB_MICRO_R);
    case LB_FRESNEL_BLEND: {
        if (!same_hemisphere(wo, wi)) return 0;
        f3 wh = normalize(wo + wi);
        float pdf_wh = mf_pdf(b.ax, b.ay, wo, wh);
        float dww = dot(wo, wh);
        return .5f * (abs_cos_theta(wi) * kInvPi + pdf_wh / (4 * dww));
    }
    default: return 0;
    }
}
LOBE_FN float lobe_pdf_v(Lobe b, float wox, float woy, float woz, float wix, float wiy, float wiz) {
    return lobe_pdf_body(b, f3(wox, woy, woz), f3(wix, wiy, wiz));
}
PD float lobe_pdf(const Lobe &b, const f3 &wo, const f3 &wi) { return lobe_pdf_v(b, wo.x, wo.y, wo.z, wi.x, wi.y, wi.z); }
PD f3 lobe_f_body(const Lobe &b, const f3 &wo, const f3 &wi) {
    switch (b.impl) {
    case LB_LAMBERT_R: case LB_LAMBERT_T: return b.S * (float)kInvPi;
    case LB_OREN_NAYAR: {
        float sinThetaI = sin_theta(wi), sinThetaO = sin_theta(wo);
        float maxCos = 0;
        if (sinThetaI > 1e-4 && sinThetaO > 1e-4) {
            float sinPhiI = sin_phi(wi), cosPhiI = cos_phi(wi);
            float sinPhiO = sin_phi(wo), cosPhiO = cos_phi(wo);
            float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
            maxCos = fmaxr(0.f, dCos);
        }
        float sinAlpha, tanBeta;
        if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sinAlpha = sinThetaO; tanBeta = sinThetaI / abs_cos_theta(wi); }
        else { sinAlpha = sinThetaI; tanBeta = sinThetaO / abs_cos_theta(wo); }
        return b.S * (float)kInvPi * (b.A + b.B * maxCos * sinAlpha * tanBeta);
    }
    case LB_MICRO_T: case LB_MICRO_R: return micro_f(b, wo, wi, b.impl == LB_MICRO_R);
    case LB_FRESNEL_BLEND: {
        f3 diffuse = (float)(28.f / (23.f * kPi)) * b.S * (f3(1.f) - b.T) * (1 - pow5(1 - .5f * abs_cos_theta(wi))) *
                     (1 - pow5(1 - .5f * abs_cos_theta(wo)));
        f3 wh = wi + wo;
        if (is_zero(wh.x) && is_zero(wh.y) && is_zero(wh.z)) return f3(0.f);
        wh = normalize(wh);
        f3 sf = b.T + pow5(1 - dot(wi, wh)) * (f3(1.f) - b.T);
        f3 specular = mf_D(b.ax, b.ay, wh) / (4 * absdot(wi, wh) * fmaxr(abs_cos_theta(wi), abs_cos_theta(wo))) * sf;
        return diffuse + specular;
    }
    case LB_BSSRDF_ADAPTER: {
        f3 f = bssrdf_sw(b.C, wi);
        f *= b.C * b.C;
        return f;
    }
    case LB_DISNEY_DIFFUSE: {
        float Fo = schlick_weight(abs_cos_theta(wo)), Fi = schlick_weight(abs_cos_theta(wi));
        return b.S * (float)kInvPi * (1 - Fo / 2) * (1 - Fi / 2);
    }
    case LB_DISNEY_FAKESS: {
        f3 wh = wi + wo;
        if (is_black(wh)) return f3(0.f);
        wh = normalize(wh);
        float cosThetaD = dot(wi, wh);
        float Fss90 = cosThetaD * cosThetaD * b.A;
        float Fo = schlick_weight(abs_cos_theta(wo)), Fi = schlick_weight(abs_cos_theta(wi));
        float Fss = ((1 - Fo) * 1.0f + Fo * Fss90) * ((1 - Fi) * 1.0f + Fi * Fss90);
        float ss = 1.25f * (Fss * (1 / (abs_cos_theta(wo) + abs_cos_theta(wi)) - .5f) + .5f);
        return b.S * (float)kInvPi * ss;
    }
    case LB_DISNEY_RETRO: {
        f3 wh = wi + wo;
        if (is_black(wh)) return f3(0.f);
        wh = normalize(wh);
        float cosThetaD = dot(wi, wh);
        float Fo = schlick_weight(abs_cos_theta(wo)), Fi = schlick_weight(abs_cos_theta(wi));
        float Rr = 2 * b.A * cosThetaD * cosThetaD;
        return b.S * (float)kInvPi * Rr * (Fo + Fi + Fo * Fi * (Rr - 1));
    }
    case LB_DISNEY_SHEEN: {
        f3 wh = wi + wo;
        if (is_black(wh)) return f3(0.f);
        wh = normalize(wh);
        float cosThetaD = absr(dot(wi, wh));
        float luminance = 0.3f * b.S.x + 0.6f * b.S.y + 1.0f * b.S.z; /* CalculateTint, reflection.cpp:272-275 */
        f3 tint = (luminance > 0) ? b.S * (1.f / luminance) : f3(1.f);
        float fSch = schlick_weight(cosThetaD);
        return b.T * ((1 - b.C) * f3(1.f) + b.C * tint) * fSch;
    }
    case LB_DISNEY_CLEARCOAT: {
        f3 wh = wi + wo;
        if (is_black(wh)) return f3(0.f);
        wh = normalize(wh);
        float Dr = gtr1(abs_cos_theta(wh), b.B);
        float w = schlick_weight(dot(wo, wh));
        float Fr = (1 - w) * .04f + w * 1.f;
        float Gr = smith_g_ggx(abs_cos_theta(wo), .25f) * smith_g_ggx(abs_cos_theta(wi), .25f);
        return f3(b.A * Gr * Fr * Dr / 4);
    }
    default: return f3(0.f);
    }
}
LOBE_FN f3 lobe_f_v(Lobe b, float wox, float woy, float woz, float wix, float wiy, float wiz) {
    return lobe_f_body(b, f3(wox, woy, woz), f3(wix, wiy, wiz));
}
PD f3 lobe_f(const Lobe &b, const f3 &wo, const f3 &wi) { return lobe_f_v(b, wo.x, wo.y, wo.z, wi.x, wi.y, wi.z); }
/* f and pdf of one lobe for the same pair of directions.  (One out-of-line function computing both, so that the half vector
 * and D are shared, was measured: its larger register footprint pushed the shade kernel from 4 to 27 spilled VGPRs and
 * made it 5 % slower, so the two switches stay separate calls.) */
struct LobeFP { f3 f; float pdf; };
PD LobeFP lobe_f_pdf(const Lobe &b, const f3 &wo, const f3 &wi) { LobeFP r; r.f = lobe_f(b, wo, wi); r.pdf = lobe_pdf(b, wo, wi); return r; }
/* by value in and out: everything travels in registers (a pointer argument would pin the caller's variable to scratch) */
struct LobeSample { f3 f, wi; float pdf; int type; };
PD f3 lobe_sample_f_body(const Lobe &b, const f3 &wo, f3 *wi, const f2 &u, float *pdf, int *sampledType);
LOBE_FN LobeSample lobe_sample_f_v(Lobe b, float wox, float woy, float woz, float ux, float uy, int type_in) {
    LobeSample r;
    r.wi = f3(0.f); r.pdf = 0; r.type = type_in;
    r.f = lobe_sample_f_body(b, f3(wox, woy, woz), &r.wi, f2(ux, uy), &r.pdf, &r.type);
    return r;
}
PD f3 lobe_sample_f(const Lobe &b, const f3 &wo, f3 *wi, const f2 &u, float *pdf, int *sampledType) {
    const LobeSample r = lobe_sample_f_v(b, wo.x, wo.y, wo.z, u.x, u.y, *sampledType);
    *wi = r.wi; *pdf = r.pdf; *sampledType = r.type;
    return r.f;
}
PD f3 lobe_sample_f_body(const Lobe &b, const f3 &wo, f3 *wi, const f2 &u, float *pdf, int *sampledType) {
    switch (b.impl) {
    case LB_LAMBERT_R: case LB_OREN_NAYAR: case LB_BSSRDF_ADAPTER:
    case LB_DISNEY_DIFFUSE: case LB_DISNEY_FAKESS: case LB_DISNEY_RETRO: case LB_DISNEY_SHEEN:
        *wi = cosine_sample_hemisphere(u);
        if (wo.z < 0) wi->z *= -1;
        *pdf = lobe_pdf(b, wo, *wi);
        return lobe_f(b, wo, *wi);
    case LB_DISNEY_CLEARCOAT: {
        if (is_zero(wo.z)) return f3(0.f);
        float alpha2 = b.B * b.B;
        float cosTheta = sqrtf(fmaxr(0.f, (1 - powf(alpha2, 1 - u.x)) / (1 - alpha2)));
        float sinTheta = sqrtf(fmaxr(0.f, 1 - cosTheta * cosTheta));
        float phi = 2 * kPi * u.y;
        f3 wh(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
        if (!same_hemisphere(wo, wh)) wh = -wh;
        *wi = reflect(wo, wh);
        if (!same_hemisphere(wo, *wi)) return f3(0.f);
        *pdf = lobe_pdf(b, wo, *wi);
        return lobe_f(b, wo, *wi);
    }
    case LB_LAMBERT_T:
        *wi = cosine_sample_hemisphere(u);
        if (wo.z > 0) wi->z *= -1;
        *pdf = lobe_pdf(b, wo, *wi);
        return lobe_f(b, wo, *wi);
    case LB_SPEC_R:
        *wi = f3(-wo.x, -wo.y, wo.z);
        *pdf = 1;
        return fresnel_eval(b, cos_theta(*wi)) * b.S / abs_cos_theta(*wi);
    case LB_SPEC_T: {
        bool entering = cos_theta(wo) > 0;
        float etaI = entering ? b.A : b.B;
        float etaT = entering ? b.B : b.A;
        if (!refract(wo, faceforward(f3(0, 0, 1), wo), etaI / etaT, wi)) return f3(0.f);
        *pdf = 1;
        f3 ft = b.S * (f3(1.f) - fresnel_eval(b, cos_theta(*wi)));
        ft *= (etaI * etaI) / (etaT * etaT);
        return ft / abs_cos_theta(*wi);
    }
    case LB_FRESNEL_SPEC: {
        float F = fr_dielectric(cos_theta(wo), b.A, b.B);
        if (u.x < F) {
            *wi = f3(-wo.x, -wo.y, wo.z);
            *sampledType = BSDF_SPECULAR | BSDF_REFLECTION;
            *pdf = F;
            return F * b.S / abs_cos_theta(*wi);
        }
        bool entering = cos_theta(wo) > 0;
        float etaI = entering ? b.A : b.B;
        float etaT = entering ? b.B : b.A;
        if (!refract(wo, faceforward(f3(0, 0, 1), wo), etaI / etaT, wi)) return f3(0.f);
        f3 ft = b.T * (1 - F);
        ft *= (etaI * etaI) / (etaT * etaT);
        *sampledType = BSDF_SPECULAR | BSDF_TRANSMISSION;
        *pdf = 1 - F;
        return ft / abs_cos_theta(*wi);
    }
    case LB_MICRO_T: case LB_MICRO_R: {
        /* one body for both (see micro_f): the half-vector sample, the density and the value are evaluated once for a wavefront
         * whose lanes picked different lobes of a rough dielectric */
        const bool refl = b.impl == LB_MICRO_R;
        if (is_zero(wo.z)) return f3(0.f);
        const f3 wh = mf_sample_wh(b.ax, b.ay, wo, u);
        if (dot(wo, wh) < 0) return f3(0.f);
        if (refl) {
            *wi = reflect(wo, wh);
            if (!same_hemisphere(wo, *wi)) return f3(0.f);
        } else {
            const float eta = cos_theta(wo) > 0 ? (b.A / b.B) : (b.B / b.A);
            if (!refract(wo, wh, eta, wi)) return f3(0.f);
        }
        /* MicrofacetReflection::Sample_f takes its density from the SAMPLED half vector (reflection.cpp), the transmission lobe
         * calls Pdf(wo, wi), which rebuilds it from the directions */
        *pdf = micro_pdf(b, wo, *wi, refl, refl ? &wh : nullptr);
        return micro_f(b, wo, *wi, refl);
    }
    case LB_FRESNEL_BLEND: {
        f2 uu = u;
        if (uu.x < .5) {
            uu.x = fminr(2 * uu.x, kOneMinusEpsilon);
            *wi = cosine_sample_hemisphere(uu);
            if (wo.z < 0) wi->z *= -1;
        } else {
            uu.x = fminr(2 * (uu.x - .5f), kOneMinusEpsilon);
            f3 wh = mf_sample_wh(b.ax, b.ay, wo, uu);
            *wi = reflect(wo, wh);
            if (!same_hemisphere(wo, *wi)) return f3(0.f);
        }
        *pdf = lobe_pdf(b, wo, *wi);
        return lobe_f(b, wo, *wi);
    }
    default: *pdf = 0; return f3(0.f);
    }
}

/* ---- BSDF, reflection.h:367-382, reflection.cpp:591-701 ---- */
PD void bsdf_init(BSDF *b, const SI &si) {
    b->n = 0;
    b->ns = si.n;
    b->ss = normalize(si.dpdu);
    b->ts = cross(b->ns, b->ss);
}
PD f3 bsdf_to_local(const BSDF &b, const f3 &v) { return f3(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }
PD f3 bsdf_to_world(const BSDF &b, const f3 &v) {
    return f3(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z, b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
PD int bsdf_num_components(const BSDF &b, int flags) {
    int num = 0;
    for (int i = 0; i < b.n; ++i) if (type_matches(lobe_type(b, i), flags)) ++num;
    return num;
}
PD f3 bsdf_f(const BSDF &b, const f3 &woW, const f3 &wiW, int flags) {
    if (b.n == 0) return f3(0.f);
    f3 wi = bsdf_to_local(b, wiW), wo = bsdf_to_local(b, woW);
    if (is_zero(wo.z)) return f3(0.f);
    bool refl = dot(wiW, b.ns) * dot(woW, b.ns) > 0;
    f3 f(0.f);
    for (int i = 0; i < b.n; ++i) {
        const int ty = lobe_type(b, i);
        if (type_matches(ty, flags) && ((refl && (ty & BSDF_REFLECTION)) || (!refl && (ty & BSDF_TRANSMISSION)))) f += lobe_f(lobe_load(b, i), wo, wi);
    }
    return f;
}
PD float bsdf_pdf(const BSDF &b, const f3 &woW, const f3 &wiW, int flags) {
    if (b.n == 0) return 0.f;
    f3 wo = bsdf_to_local(b, woW), wi = bsdf_to_local(b, wiW);
    if (is_zero(wo.z)) return 0.;
    float pdf = 0.f;
    int matching = 0;
    for (int i = 0; i < b.n; i++)
        if (type_matches(lobe_type(b, i), flags)) { pdf += lobe_pdf(lobe_load(b, i), wo, wi); matching += 1; }
    return matching > 0 ? pdf / matching : 0.f;
}
/* BSDF::f and BSDF::Pdf for the same directions and flags (EstimateDirect asks for both, primitive.cpp:262-264) */
PD f3 bsdf_f_pdf(const BSDF &b, const f3 &woW, const f3 &wiW, int flags, float *pdf_out) {
    *pdf_out = 0.f;
    if (b.n == 0) return f3(0.f);
    f3 wi = bsdf_to_local(b, wiW), wo = bsdf_to_local(b, woW);
    if (is_zero(wo.z)) return f3(0.f);
    bool refl = dot(wiW, b.ns) * dot(woW, b.ns) > 0;
    f3 f(0.f);
    float pdf = 0.f;
    int matching = 0;
    for (int i = 0; i < b.n; ++i) {
        const int ty = lobe_type(b, i);
        if (!type_matches(ty, flags)) continue;
        matching += 1;
        if ((refl && (ty & BSDF_REFLECTION)) || (!refl && (ty & BSDF_TRANSMISSION))) {
            const LobeFP r = lobe_f_pdf(lobe_load(b, i), wo, wi);
            f += r.f;
            pdf += r.pdf;
        } else pdf += lobe_pdf(lobe_load(b, i), wo, wi);
    }
    *pdf_out = matching > 0 ? pdf / matching : 0.f;
    return f;
}
PD f3 bsdf_sample_f(const BSDF &b, const f3 &woW, f3 *wiW, const f2 &u, float *pdf, int type, int *sampledType) {
    int matching = bsdf_num_components(b, type);
    if (matching == 0) { *pdf = 0; *sampledType = 0; return f3(0.f); }
    int comp_i = (int)fminr((float)(int)floorf(u.x * matching), (float)(matching - 1));
    int count = comp_i, picked = -1;
    for (int i = 0; i < b.n; ++i)
        if (type_matches(lobe_type(b, i), type) && count-- == 0) { picked = i; break; }
    f2 uR(fminr(u.x * matching - comp_i, kOneMinusEpsilon), u.y);
    f3 wi, wo = bsdf_to_local(b, woW);
    if (is_zero(wo.z)) return f3(0.f);
    *pdf = 0;
    int ptype = lobe_type(b, picked);
    *sampledType = ptype;
    f3 f = lobe_sample_f(lobe_load(b, picked), wo, &wi, uR, pdf, sampledType);
    if (is_zero(*pdf)) { *sampledType = 0; return f3(0.f); }
    *wiW = bsdf_to_world(b, wi);
    /* reflection.cpp:626-648: the other matching lobes add their density, and f is summed again over all matching lobes
     * on the side (wo, wi) are on.  The sampled lobe's own term is the value its Sample_f returned - BxDF::f of the same
     * lobe for the same directions - and a lobe that contributes both f and pdf is evaluated once for both. */
    if (!(ptype & BSDF_SPECULAR)) {
        const bool refl = dot(*wiW, b.ns) * dot(woW, b.ns) > 0;
        const f3 f_picked = f;
        f = f3(0.f);
        for (int i = 0; i < b.n; ++i) {
            const int ty = lobe_type(b, i);
            if (!type_matches(ty, type)) continue;
            const bool side = (refl && (ty & BSDF_REFLECTION)) || (!refl && (ty & BSDF_TRANSMISSION));
            if (i == picked) {
                if (side) f += f_picked;
            } else if (side) {
                const LobeFP r = lobe_f_pdf(lobe_load(b, i), wo, wi);
                f += r.f;
                *pdf += r.pdf;
            } else *pdf += lobe_pdf(lobe_load(b, i), wo, wi);
        }
    }
    if (matching > 1) *pdf /= matching;
    return f;
}

/* ---- textures, texture.h:18-84, scene.cpp:769-788 ---- */
PD float inverse_gamma(float value) {
    if (value <= 0.04045f) return value * 1.f / 12.92f;
    return powf((value + 0.055f) * 1.f / 1.055f, 2.4f);
}
PD f3 tex_image(const DScene &sc, const DTex &t, const f2 &uv) {
    const DImage &im = sc.images[t.image];
    int i = uv.x * im.width, j = uv.y * im.height;
    if (i < 0) i = 0;
    if (i > im.width - 1) i = im.width - 1;
    if (j < 0) j = 0;
    if (j > im.height - 1) j = im.height - 1;
    const float *p = im.rgb + 3 * (size_t)(i + im.width * j);
    return f3(1.f * inverse_gamma(p[0]), 1.f * inverse_gamma(p[1]), 1.f * inverse_gamma(p[2]));
}
PD f3 tex_spec(const DScene &sc, const DTex &t, const f2 &uv) {
    if (t.type == 0) return f3(t.v[0], t.v[1], t.v[2]);
    if (t.type == 1) return f3(t.v[0]);
    return tex_image(sc, t, uv);
}
PD float tex_float(const DScene &sc, const DTex &t, const f2 &uv) {
    if (t.type == 2) return tex_image(sc, t, uv).x;
    return t.v[0];
}

/* ---- lobe constructors, reflection.h:131-246 ---- */
PD void lobe_clear(Lobe &b) { /* fields a lobe kind does not set are never read by it */
    b.type = 0; b.impl = LB_INVALID; b.fr_type = FR_NOOP;
    b.S = f3(0.f); b.T = f3(0.f); b.A = b.B = b.C = 0; b.ax = b.ay = 0; b.fr_etaI = b.fr_etaT = 0;
}
PD void push_lambert_r(BSDF *b, const f3 &R) { Lobe l; lobe_clear(l); l.type = BSDF_REFLECTION | BSDF_DIFFUSE; l.impl = LB_LAMBERT_R; l.S = R; lobe_push(b, l); }
PD void push_lambert_t(BSDF *b, const f3 &T) { Lobe l; lobe_clear(l); l.type = BSDF_TRANSMISSION | BSDF_DIFFUSE; l.impl = LB_LAMBERT_T; l.S = T; lobe_push(b, l); }
PD void push_oren_nayar(BSDF *b, const f3 &R, float sigma) {
    Lobe l; lobe_clear(l); l.type = BSDF_REFLECTION | BSDF_DIFFUSE; l.impl = LB_OREN_NAYAR; l.S = R;
    float sigma2 = sigma * sigma;
    l.A = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
    l.B = 0.45f * sigma2 / (sigma2 + 0.09f);
    lobe_push(b, l);
}
PD void push_spec_r(BSDF *b, const f3 &R, int frt, float eI, float eT) {
    Lobe l; lobe_clear(l); l.type = BSDF_REFLECTION | BSDF_SPECULAR; l.impl = LB_SPEC_R; l.S = R;
    l.fr_type = frt; l.fr_etaI = eI; l.fr_etaT = eT;
    lobe_push(b, l);
}
PD void push_spec_t(BSDF *b, const f3 &T, float eA, float eB) {
    Lobe l; lobe_clear(l); l.type = BSDF_TRANSMISSION | BSDF_SPECULAR; l.impl = LB_SPEC_T; l.S = T; l.A = eA; l.B = eB;
    l.fr_type = FR_DIELECTRIC; l.fr_etaI = eA; l.fr_etaT = eB;
    lobe_push(b, l);
}
PD void push_fresnel_spec(BSDF *b, const f3 &R, const f3 &T, float eA, float eB) {
    Lobe l; lobe_clear(l); l.type = BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR; l.impl = LB_FRESNEL_SPEC;
    l.S = R; l.T = T; l.A = eA; l.B = eB;
    lobe_push(b, l);
}
PD void push_micro_t(BSDF *b, const f3 &T, float eA, float eB, float ax, float ay, bool remap) {
    Lobe l; lobe_clear(l); l.type = BSDF_TRANSMISSION | BSDF_GLOSSY; l.impl = LB_MICRO_T; l.A = eA; l.B = eB; l.T = T;
    l.fr_type = FR_DIELECTRIC; l.fr_etaI = eA; l.fr_etaT = eB;
    l.ax = remap ? roughness_to_alpha(ax) : ax;
    l.ay = remap ? roughness_to_alpha(ay) : ay;
    lobe_push(b, l);
}
PD void push_micro_r_diel(BSDF *b, const f3 &R, float ax, float ay, float eI, float eT, bool remap) {
    Lobe l; lobe_clear(l); l.type = BSDF_REFLECTION | BSDF_GLOSSY; l.impl = LB_MICRO_R; l.S = R;
    l.fr_type = FR_DIELECTRIC; l.fr_etaI = eI; l.fr_etaT = eT;
    l.ax = remap ? roughness_to_alpha(ax) : ax;
    l.ay = remap ? roughness_to_alpha(ay) : ay;
    lobe_push(b, l);
}
/* glass.cpp:26-57 and the glossy halves of subsurface.cpp:35-67 / kdsubsurface.cpp:126-158 */
PD void glass_lobes(BSDF *b, const f3 &R, const f3 &T, float eta, float ur, float vr, bool subsurfaceOrder) {
    bool isSpecular = is_zero(ur) && is_zero(vr);
    if (isSpecular) { push_fresnel_spec(b, R, T, 1.f, eta); return; }
    if (subsurfaceOrder) {
        if (!is_black(R)) push_micro_r_diel(b, R, ur, vr, 1.f, eta, false);
        if (!is_black(T)) push_micro_t(b, T, 1.f, eta, ur, vr, false);
    } else {
        if (!is_black(T)) push_micro_t(b, T, 1.f, eta, ur, vr, true);
        if (!is_black(R)) push_micro_r_diel(b, R, ur, vr, 1.f, eta, true);
    }
}
PD f3 clamp11(const f3 &v) { return f3(clampr(v.x, -1, 1), clampr(v.y, -1, 1), clampr(v.z, -1, 1)); }

/* Subsurface parameters produced alongside the lobes for the two SSS materials */
struct BssrdfParams {
    int has;
    float eta;
    f3 sigma_a, sigma_s;
    int table;
    f3 kd, mfree; /* KdSubsurface: inverted on demand (SubsurfaceFromDiffuse) */
    int kd_mode;
};

/* Primitive::ComputeScatteringFunctions -> Material switch -> per-material builders
 * (primitive.cpp:89-106, material.cpp:103-165, materials/).  Returns false for an emitter
 * (whose single invalidated lobe matches nothing: the path ends there). */
PD bool compute_scattering(const DScene &sc, const SI &si, BSDF *bsdf, BssrdfParams *bp) {
    const DPrim &pr = sc.prims[si.prim];
    bp->has = 0;
    if (pr.flags & PF_EMISSIVE) return false;
    if (pr.material < 0) return true;
    const DMat &m = sc.mats[pr.material];
    const DTex *tx = m.tex;
    const f2 uv = si.uv;
    switch (m.type) {
    case MT_MATTE: {
        f3 kd = tex_spec(sc, tx[0], uv);
        float sig = tex_float(sc, tx[1], uv);
        if (!is_black(kd)) {
            if (is_zero(sig)) push_lambert_r(bsdf, kd);
            else push_oren_nayar(bsdf, kd, sig);
        }
    } break;
    case MT_MIRROR: {
        f3 kr = tex_spec(sc, tx[0], uv);
        if (!is_black(kr)) push_spec_r(bsdf, kr, FR_NOOP, 0, 0);
    } break;
    case MT_GLASS: {
        float eta = tex_float(sc, tx[4], uv);
        float ur = tex_float(sc, tx[2], uv), vr = tex_float(sc, tx[3], uv);
        f3 R = tex_spec(sc, tx[0], uv), T = tex_spec(sc, tx[1], uv);
        if (is_black(R) && is_black(T)) break;
        glass_lobes(bsdf, R, T, eta, ur, vr, false);
    } break;
    case MT_METAL: {
        float ur = tex_float(sc, tx[2], uv), vr = tex_float(sc, tx[3], uv);
        Lobe l;
        lobe_clear(l);
        l.type = BSDF_REFLECTION | BSDF_GLOSSY; l.impl = LB_MICRO_R; l.S = f3(1.f);
        l.fr_type = FR_CONDUCTOR; l.T = tex_spec(sc, tx[0], uv);
        { f3 k = tex_spec(sc, tx[1], uv); l.A = k.x; l.B = k.y; l.C = k.z; }
        l.ax = roughness_to_alpha(ur); l.ay = roughness_to_alpha(vr);
        lobe_push(bsdf, l);
    } break;
    case MT_PLASTIC: {
        f3 kd = tex_spec(sc, tx[0], uv);
        if (!is_black(kd)) push_lambert_r(bsdf, kd);
        f3 ks = tex_spec(sc, tx[1], uv);
        if (!is_black(ks)) {
            float rough = tex_float(sc, tx[2], uv);
            push_micro_r_diel(bsdf, ks, rough, rough, 1.5f, 1.f, true);
        }
    } break;
    case MT_UBER: {
        float e = tex_float(sc, tx[7], uv);
        f3 op = tex_spec(sc, tx[6], uv);
        f3 t = (-op + f3(1.f));
        if (!is_black(t)) push_spec_t(bsdf, t, 1.f, 1.f);
        f3 kd = op * tex_spec(sc, tx[0], uv);
        if (!is_black(kd)) push_lambert_r(bsdf, kd);
        f3 ks = op * tex_spec(sc, tx[1], uv);
        if (!is_black(ks)) {
            float ru = tex_float(sc, tx[4], uv), rv = tex_float(sc, tx[5], uv);
            push_micro_r_diel(bsdf, ks, ru, rv, 1.f, e, true);
        }
        f3 kr = op * tex_spec(sc, tx[2], uv);
        if (!is_black(kr)) push_spec_r(bsdf, kr, FR_DIELECTRIC, 1.f, e);
        f3 kt = op * tex_spec(sc, tx[3], uv);
        if (!is_black(kt)) push_spec_t(bsdf, kt, 1.f, e);
    } break;
    case MT_TRANSLUCENT: {
        float eta = 1.5f;
        f3 r = tex_spec(sc, tx[2], uv), t = tex_spec(sc, tx[3], uv);
        if (is_black(r) && is_black(t)) break;
        f3 kd = tex_spec(sc, tx[0], uv);
        if (!is_black(kd)) {
            if (!is_black(r)) push_lambert_r(bsdf, r * kd);
            if (!is_black(t)) push_lambert_t(bsdf, t * kd);
        }
        f3 ks = tex_spec(sc, tx[1], uv);
        if (!is_black(ks) && (!is_black(r) || !is_black(t))) {
            float rough = tex_float(sc, tx[4], uv);
            if (!is_black(r)) push_micro_r_diel(bsdf, r * ks, rough, rough, 1.f, eta, true);
            if (!is_black(t)) push_micro_t(bsdf, t * ks, 1.f, eta, rough, rough, true);
        }
    } break;
    case MT_SUBSURFACE: {
        float eta = m.vals[1], scale = m.vals[0];
        f3 R = clamp11(tex_spec(sc, tx[0], uv)), T = clamp11(tex_spec(sc, tx[1], uv));
        float ur = tex_float(sc, tx[4], uv), vr = tex_float(sc, tx[5], uv);
        if (is_black(R) && is_black(T)) break;
        glass_lobes(bsdf, R, T, eta, ur, vr, true);
        bp->has = 1; bp->eta = eta; bp->table = m.table; bp->kd_mode = 0;
        bp->sigma_a = scale * tex_spec(sc, tx[2], uv);
        bp->sigma_s = scale * tex_spec(sc, tx[3], uv);
    } break;
    case MT_KDSUBSURFACE: {
        float eta = m.vals[0], scale = m.vals[1];
        f3 R = clamp11(tex_spec(sc, tx[1], uv)), T = clamp11(tex_spec(sc, tx[2], uv));
        float ur = tex_float(sc, tx[4], uv), vr = tex_float(sc, tx[5], uv);
        if (is_black(R) && is_black(T)) break;
        glass_lobes(bsdf, R, T, eta, ur, vr, true);
        bp->has = 1; bp->eta = eta; bp->table = m.table; bp->kd_mode = 1;
        bp->kd = tex_spec(sc, tx[0], uv);
        bp->mfree = scale * tex_spec(sc, tx[3], uv);
    } break;
    case MT_DISNEY: { /* the reference's five Disney lobes assembled as PBRT-v3's DisneyMaterial assembles its own (no metallic /
                         transmission / specular terms: the reference has no lobes for them).  Slots: 0 colour, 1 roughness,
                         2 sheen, 3 sheenTint, 4 clearcoat, 5 clearcoatGloss, 6 flatness */
        f3 c = tex_spec(sc, tx[0], uv);
        float rough = tex_float(sc, tx[1], uv), sheen = tex_float(sc, tx[2], uv), stint = tex_float(sc, tx[3], uv);
        float cc = tex_float(sc, tx[4], uv), gloss = tex_float(sc, tx[5], uv), flat = tex_float(sc, tx[6], uv);
        Lobe l;
        if (!is_black(c)) {
            lobe_clear(l); l.type = BSDF_REFLECTION | BSDF_DIFFUSE; l.impl = LB_DISNEY_DIFFUSE; l.S = (1 - flat) * c; lobe_push(bsdf, l);
            if (flat > 0) { lobe_clear(l); l.type = BSDF_REFLECTION | BSDF_DIFFUSE; l.impl = LB_DISNEY_FAKESS; l.S = flat * c; l.A = rough; lobe_push(bsdf, l); }
            lobe_clear(l); l.type = BSDF_REFLECTION | BSDF_DIFFUSE; l.impl = LB_DISNEY_RETRO; l.S = c; l.A = rough; lobe_push(bsdf, l);
            if (sheen > 0) { lobe_clear(l); l.type = BSDF_REFLECTION | BSDF_DIFFUSE; l.impl = LB_DISNEY_SHEEN; l.S = c; l.T = f3(sheen); l.C = stint; lobe_push(bsdf, l); }
        }
        if (cc > 0) { /* Init_DisneyClearcoat (reflection.h:90-95) with its two numbers in the fields the lobe reads (A, B) */
            lobe_clear(l); l.type = BSDF_REFLECTION | BSDF_GLOSSY; l.impl = LB_DISNEY_CLEARCOAT; l.A = cc; l.B = (1 - gloss) * .1f + gloss * .001f; lobe_push(bsdf, l);
        }
    } break;
    case MT_PLAYGROUND: {
        f3 base = tex_spec(sc, tx[0], uv);
        for (int i = 0; i < m.ntex; i++)
            if (si.face == i) { base = tex_spec(sc, tx[i], uv); break; }
        push_lambert_r(bsdf, base);
    } break;
    default: break;
    }
    return true;
}
