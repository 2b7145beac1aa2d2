/*
 * orc_bsdf.cpp — ORACLE (test infrastructure only).  PARITY UNPINNED.
 * Restates: Fresnel (src/core/material.cpp:5-86), Trowbridge-Reitz microfacets
 * (src/core/microfacet.h:30-43, microfacet.cpp:3-120), BxDF lobes and BSDF
 * (src/core/reflection.cpp:36-701, reflection.h:358-406), textures (texture.h:18-84) and the
 * material -> BSDF builders (src/materials/*.cpp, src/core/primitive.cpp:89-106).
 * Disney lobes are omitted: no material instantiates them.
 */
#include "orc_scene.h"

namespace orc {

/* ---- material.cpp:5-52 -------------------------------------------------------------------- */
Float FrDieletric(Float cosThetaI, Float etaI, Float etaT) {
    cosThetaI = Clampf(cosThetaI, -1, 1);
    bool entering = cosThetaI > 0.f;
    if (!entering) {
        Float s = etaI; etaI = etaT; etaT = s;
        cosThetaI = Absf(cosThetaI);
    }
    Float sinThetaI = sqrtf(Max((Float)0, 1 - cosThetaI * cosThetaI));
    Float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    Float cosThetaT = sqrtf(Max((Float)0, 1 - sinThetaT * sinThetaT));
    Float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    Float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}
V3 FrConductor(Float cosThetaI, const V3 &etai, const V3 &etat, const V3 &k) {
    cosThetaI = Clampf(cosThetaI, -1, 1);
    V3 eta = divv(etat, etai);
    V3 etak = divv(k, etai);
    Float cosThetaI2 = cosThetaI * cosThetaI;
    Float sinThetaI2 = 1. - cosThetaI2;
    V3 eta2 = eta * eta;
    V3 etak2 = etak * etak;
    V3 t0 = eta2 - etak2 - V3(sinThetaI2);
    V3 a2plusb2 = SqrtV(t0 * t0 + 4 * eta2 * etak2);
    V3 t1 = a2plusb2 + V3(cosThetaI2);
    V3 a = SqrtV(0.5f * (a2plusb2 + t0));
    V3 t2 = (Float)2 * cosThetaI * a;
    V3 Rs = divv(t1 - t2, t1 + t2);
    V3 t3 = cosThetaI2 * a2plusb2 + V3(sinThetaI2 * sinThetaI2);
    V3 t4 = t2 * sinThetaI2;
    V3 Rp = divv(Rs * (t3 - t4), t3 + t4);
    return 0.5 * (Rp + Rs);
}
static V3 FresnelEvaluate(const Fresnel &fr, Float cosI) { /* material.cpp:54-73 */
    switch (fr.type) {
    case FrDielectric: return V3(FrDieletric(cosI, fr.etaI, fr.etaT));
    case FrConductorT: return FrConductor(Absf(cosI), fr.setaI, fr.setaT, fr.sk);
    default: return V3(1.f);
    }
}

/* ---- microfacet.h / microfacet.cpp --------------------------------------------------------- */
Float RoughnessToAlpha(Float roughness) {
    roughness = Max(roughness, (Float)1e-3);
    Float x = logf(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
static Float MF_Lambda(const MicrofacetDistribution &m, const V3 &w) {
    Float absTanTheta = Absf(TanTheta(w));
    if (std::isinf(absTanTheta)) return 0.;
    Float alpha = sqrtf(Cos2Phi(w) * m.alphax * m.alphax + Sin2Phi(w) * m.alphay * m.alphay);
    Float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
    return (-1 + sqrtf(1.f + alpha2Tan2Theta)) / 2;
}
static Float MF_D(const MicrofacetDistribution &m, const V3 &wh) {
    Float tan2Theta = Tan2Theta(wh);
    if (std::isinf(tan2Theta)) return 0.;
    const Float cos4Theta = Cos2Theta(wh) * Cos2Theta(wh);
    Float e = (Cos2Phi(wh) / (m.alphax * m.alphax) + Sin2Phi(wh) / (m.alphay * m.alphay)) * tan2Theta;
    return 1 / (ORC_Pi * m.alphax * m.alphay * cos4Theta * (1 + e) * (1 + e));
}
static Float MF_G1(const MicrofacetDistribution &m, const V3 &w) { return 1 / (1 + MF_Lambda(m, w)); }
static Float MF_G(const MicrofacetDistribution &m, const V3 &wo, const V3 &wi) { return 1 / (1 + MF_Lambda(m, wo) + MF_Lambda(m, wi)); }
static Float MF_Pdf(const MicrofacetDistribution &m, const V3 &wo, const V3 &wh) {
    if (m.sampleVisibleArea) return MF_D(m, wh) * MF_G1(m, wo) * AbsDot(wo, wh) / AbsCosTheta(wo);
    return MF_D(m, wh) * AbsCosTheta(wh);
}
static void Sample11(Float cosTheta, Float U1, Float U2, Float *slope_x, Float *slope_y) { /* microfacet.cpp:23-61 */
    if (cosTheta > .9999) {
        Float r = sqrtf(U1 / (1 - U1));
        Float phi = 6.28318530718 * U2;
        *slope_x = r * cosf(phi); /* cos(Float) resolves to the float overload */
        *slope_y = r * sinf(phi);
        return;
    }
    Float sinTheta = sqrtf(Max((Float)0, (Float)1 - cosTheta * cosTheta));
    Float tanTheta = sinTheta / cosTheta;
    Float a = 1 / tanTheta;
    Float G1 = 2 / (1 + sqrtf(1.f + 1.f / (a * a)));
    Float A = 2 * U1 / G1 - 1;
    Float tmp = 1.f / (A * A - 1.f);
    if (tmp > 1e10) tmp = 1e10;
    Float B = tanTheta;
    Float D = sqrtf(Max(Float(B * B * tmp * tmp - (A * A - B * B) * tmp), Float(0)));
    Float slope_x_1 = B * tmp - D;
    Float slope_x_2 = B * tmp + D;
    *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
    Float S;
    if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
    else { S = -1.f; U2 = 2.f * (.5f - U2); }
    Float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
              (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    *slope_y = S * z * sqrtf(1.f + *slope_x * *slope_x);
}
static V3 MF_Sample_wh(const MicrofacetDistribution &m, const V3 &wo, const V2 &u) { /* microfacet.cpp:63-120 */
    V3 wh;
    if (!m.sampleVisibleArea) {
        Float cosTheta = 0, phi = (2 * ORC_Pi) * u[1];
        if (m.alphax == m.alphay) {
            Float tanTheta2 = m.alphax * m.alphax * u[0] / (1.0f - u[0]);
            cosTheta = 1 / sqrtf(1 + tanTheta2);
        } else {
            phi = atanf(m.alphay / m.alphax * tanf(2 * ORC_Pi * u[1] + .5f * ORC_Pi));
            if (u[1] > .5f) phi += ORC_Pi;
            Float sinPhi = sinf(phi), cosPhi = cosf(phi);
            const Float alphax2 = m.alphax * m.alphax, alphay2 = m.alphay * m.alphay;
            const Float alpha2 = 1 / (cosPhi * cosPhi / alphax2 + sinPhi * sinPhi / alphay2);
            Float tanTheta2 = alpha2 * u[0] / (1 - u[0]);
            cosTheta = 1 / sqrtf(1 + tanTheta2);
        }
        Float sinTheta = sqrtf(Max((Float)0., (Float)1. - cosTheta * cosTheta));
        wh = SphericalDirection(sinTheta, cosTheta, phi);
        if (!SameHemisphere(wo, wh)) wh = -wh;
    } else {
        bool flip = wo.z < 0;
        V3 wi = flip ? -wo : wo;
        V3 wiStretched = Normalize(V3(m.alphax * wi.x, m.alphay * wi.y, wi.z));
        Float slope_x, slope_y;
        Sample11(CosTheta(wiStretched), u[0], u[1], &slope_x, &slope_y);
        Float tmp = CosPhi(wiStretched) * slope_x - SinPhi(wiStretched) * slope_y;
        slope_y = SinPhi(wiStretched) * slope_x + CosPhi(wiStretched) * slope_y;
        slope_x = tmp;
        slope_x = m.alphax * slope_x;
        slope_y = m.alphay * slope_y;
        wh = Normalize(V3(-slope_x, -slope_y, 1.));
        if (flip) wh = -wh;
    }
    return wh;
}

/* ---- reflection.cpp lobes -------------------------------------------------------------------- */
static Float pow5(Float v) { return v * v * v * v * v; }
static V3 SchlickFresnel(Float cosTheta, const V3 &Rs) { return Rs + pow5(1 - cosTheta) * (V3(1.) - Rs); }

/* ---- Disney helpers, reflection.cpp:3-26 ---- */
static Float SchlickWeight(Float cosTheta) {
    Float m = Clampf(1 - cosTheta, 0, 1);
    return (m * m) * (m * m) * m;
}
static Float FrSchlick(Float R0, Float cosTheta) { Float t = SchlickWeight(cosTheta); return (1 - t) * R0 + t * 1.f; } /* Lerp(t, R0, 1), geometry.h:1025 */
static Float GTR1(Float cosTheta, Float alpha) {
    if (alpha > 1) return ORC_InvPi;
    Float alpha2 = alpha * alpha;
    return (alpha2 - 1) / (ORC_Pi * std::log(alpha2) * (1 + (alpha2 - 1) * cosTheta * cosTheta));
}
static Float smithG_GGX(Float cosTheta, Float alpha) {
    Float alpha2 = alpha * alpha;
    Float cosTheta2 = cosTheta * cosTheta;
    return 2.f / (1.f + std::sqrt(alpha2 + (1.0f - alpha2) * cosTheta2));
}
static Float DisneyClearcoat_Pdf(const BxDF &b, const V3 &wo, const V3 &wi) { /* :77-84 */
    if (!SameHemisphere(wo, wi)) return 0;
    V3 wh = wi + wo;
    if (IsBlack(wh)) return 0;
    wh = Normalize(wh);
    Float Dr = GTR1(AbsCosTheta(wh), b.B);
    return Dr * AbsCosTheta(wh) / (4 * Dot(wo, wh));
}

Float BxDF_Pdf(const BxDF &b, const V3 &wo, const V3 &wi) { /* :36-147 */
    if (!b.is_valid) return 0;
    switch (b.impl) {
    case LambertianReflection: case OrenNayar: case BSSRDFAdapter:
    case DisneySheen: case DisneyRetro: case DisneyFakeSS: case DisneyDiffuse: /* :128-133 */
        return SameHemisphere(wo, wi) ? AbsCosTheta(wi) * ORC_InvPi : 0;
    case DisneyClearcoat: return DisneyClearcoat_Pdf(b, wo, wi);
    case LambertianTransmission: return !SameHemisphere(wo, wi) ? AbsCosTheta(wi) * ORC_InvPi : 0;
    case MicrofacetTransmission: {
        if (SameHemisphere(wo, wi)) return 0;
        Float eta = CosTheta(wo) > 0 ? (b.B / b.A) : (b.A / b.B);
        V3 h = wo + wi * eta;
        V3 wh = Normalize(h);
        Float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
        Float dwh_dwi = Absf((eta * eta * Dot(wi, wh)) / (sqrtDenom * sqrtDenom));
        return MF_Pdf(b.mDist, wo, wh) * dwh_dwi;
    }
    case MicrofacetReflection: {
        if (!SameHemisphere(wo, wi)) return 0;
        V3 wh = Normalize(wo + wi);
        Float dww = Dot(wo, wh);
        return MF_Pdf(b.mDist, wo, wh) / (4 * dww);
    }
    case FresnelBlend: {
        if (!SameHemisphere(wo, wi)) return 0;
        V3 wh = Normalize(wo + wi);
        Float pdf_wh = MF_Pdf(b.mDist, wo, wh);
        Float dww = Dot(wo, wh);
        return .5f * (AbsCosTheta(wi) * ORC_InvPi + pdf_wh / (4 * dww));
    }
    default: return 0; /* specular lobes */
    }
}

V3 BxDF_f(const BxDF &b, const V3 &wo, const V3 &wi) { /* :149-367 */
    if (!b.is_valid) return V3(0);
    switch (b.impl) {
    case LambertianReflection: case LambertianTransmission: return b.S * ORC_InvPi;
    case OrenNayar: {
        Float sinThetaI = SinTheta(wi), sinThetaO = SinTheta(wo);
        Float maxCos = 0;
        if (sinThetaI > 1e-4 && sinThetaO > 1e-4) {
            Float sinPhiI = SinPhi(wi), cosPhiI = CosPhi(wi);
            Float sinPhiO = SinPhi(wo), cosPhiO = CosPhi(wo);
            Float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
            maxCos = Max((Float)0, dCos);
        }
        Float sinAlpha, tanBeta;
        if (AbsCosTheta(wi) > AbsCosTheta(wo)) { sinAlpha = sinThetaO; tanBeta = sinThetaI / AbsCosTheta(wi); }
        else { sinAlpha = sinThetaI; tanBeta = sinThetaO / AbsCosTheta(wo); }
        return b.S * ORC_InvPi * (b.A + b.B * maxCos * sinAlpha * tanBeta);
    }
    case MicrofacetTransmission: {
        if (SameHemisphere(wo, wi)) return V3(0);
        Float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
        if (IsZero(cosThetaI) || IsZero(cosThetaO)) return V3(0);
        Float eta = CosTheta(wo) > 0 ? (b.B / b.A) : (b.A / b.B);
        V3 wh = Normalize(wo + wi * eta);
        if (wh.z < 0) wh = -wh;
        V3 F = FresnelEvaluate(b.fresnel, Dot(wo, wh));
        Float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
        Float factor = 1 / eta; /* mode == Radiance always */
        return (V3(1.f) - F) * b.T *
               Absf(MF_D(b.mDist, wh) * MF_G(b.mDist, wo, wi) * eta * eta * AbsDot(wi, wh) * AbsDot(wo, wh) * factor * factor /
                    (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
    }
    case MicrofacetReflection: {
        Float cosThetaO = AbsCosTheta(wo), cosThetaI = AbsCosTheta(wi);
        V3 wh = wi + wo;
        if (IsZero(cosThetaI) || IsZero(cosThetaO)) return V3(0.);
        if (IsZero(wh.x) && IsZero(wh.y) && IsZero(wh.z)) return V3(0.);
        wh = Normalize(wh);
        V3 F = FresnelEvaluate(b.fresnel, Dot(wi, Faceforward(wh, V3(0, 0, 1))));
        return b.S * MF_D(b.mDist, wh) * MF_G(b.mDist, wo, wi) * F / (4 * cosThetaI * cosThetaO);
    }
    case FresnelBlend: {
        V3 diffuse = (Float)(28.f / (23.f * ORC_Pi)) * b.S * (V3(1.f) - b.T) * (1 - pow5(1 - .5f * AbsCosTheta(wi))) *
                     (1 - pow5(1 - .5f * AbsCosTheta(wo)));
        V3 wh = wi + wo;
        if (IsZero(wh.x) && IsZero(wh.y) && IsZero(wh.z)) return V3(0);
        wh = Normalize(wh);
        V3 specular = MF_D(b.mDist, wh) / (4 * AbsDot(wi, wh) * Max(AbsCosTheta(wi), AbsCosTheta(wo))) *
                      SchlickFresnel(Dot(wi, wh), b.T);
        return diffuse + specular;
    }
    case BSSRDFAdapter: { /* :225-230 */
        V3 f = BSSRDF_Sw(*b.bssrdf, wi);
        f *= b.bssrdf->eta * b.bssrdf->eta;
        return f;
    }
    case DisneyDiffuse: { /* :232-239 */
        Float Fo = SchlickWeight(AbsCosTheta(wo));
        Float Fi = SchlickWeight(AbsCosTheta(wi));
        return b.S * ORC_InvPi * (1 - Fo / 2) * (1 - Fi / 2);
    }
    case DisneyFakeSS: { /* :241-256 */
        V3 wh = wi + wo;
        if (IsBlack(wh)) return V3(0.);
        wh = Normalize(wh);
        Float cosThetaD = Dot(wi, wh);
        Float Fss90 = cosThetaD * cosThetaD * b.A;
        Float Fo = SchlickWeight(AbsCosTheta(wo)), Fi = SchlickWeight(AbsCosTheta(wi));
        Float Fss = ((1 - Fo) * 1.0f + Fo * Fss90) * ((1 - Fi) * 1.0f + Fi * Fss90); /* Lerp(Fo, 1, Fss90) * Lerp(Fi, 1, Fss90) */
        Float ss = 1.25f * (Fss * (1 / (AbsCosTheta(wo) + AbsCosTheta(wi)) - .5f) + .5f);
        return b.S * ORC_InvPi * ss;
    }
    case DisneyRetro: { /* :258-270 */
        V3 wh = wi + wo;
        if (IsBlack(wh)) return V3(0.);
        wh = Normalize(wh);
        Float cosThetaD = Dot(wi, wh);
        Float Fo = SchlickWeight(AbsCosTheta(wo));
        Float Fi = SchlickWeight(AbsCosTheta(wi));
        Float Rr = 2 * b.A * cosThetaD * cosThetaD;
        return b.S * ORC_InvPi * Rr * (Fo + Fi + Fo * Fi * (Rr - 1));
    }
    case DisneySheen: { /* :272-285 */
        V3 wh = wi + wo;
        if (IsBlack(wh)) return V3(0.);
        wh = Normalize(wh);
        Float cosThetaD = Absf(Dot(wi, wh));
        Float luminance = Dot(V3(0.3f, 0.6f, 1.0f), b.S);
        V3 tint = (luminance > 0) ? b.S * (1.f / luminance) : V3(1);
        Float fSch = SchlickWeight(cosThetaD);
        return b.sheen * ((1 - b.sheenTint) * V3(1) + b.sheenTint * tint) * fSch;
    }
    case DisneyClearcoat: { /* :287-295 */
        V3 wh = wi + wo;
        if (IsBlack(wh)) return V3(0.);
        wh = Normalize(wh);
        Float Dr = GTR1(AbsCosTheta(wh), b.B);
        Float Fr = FrSchlick(.04, Dot(wo, wh));
        Float Gr = smithG_GGX(AbsCosTheta(wo), .25) * smithG_GGX(AbsCosTheta(wi), .25);
        return V3(b.A * Gr * Fr * Dr / 4);
    }
    default: return V3(0.f);
    }
}

V3 BxDF_Sample_f(const BxDF &b, const V3 &wo, V3 *wi, const V2 &u, Float *pdf, int *sampledType) { /* :370-589 */
    if (!b.is_valid) { *pdf = 0; return V3(0); }
    switch (b.impl) {
    case LambertianReflection: case OrenNayar: case BSSRDFAdapter:
    case DisneySheen: case DisneyRetro: case DisneyFakeSS: case DisneyDiffuse: /* :568-574 */
        *wi = CosineSampleHemisphere(u);
        if (wo.z < 0) wi->z *= -1;
        *pdf = BxDF_Pdf(b, wo, *wi);
        return BxDF_f(b, wo, *wi);
    case DisneyClearcoat: { /* :504-521 */
        if (IsZero(wo.z)) return V3(0.);
        Float alpha2 = b.B * b.B;
        Float cosTheta = std::sqrt(Max(Float(0), (1 - std::pow(alpha2, 1 - u.x)) / (1 - alpha2)));
        Float sinTheta = std::sqrt(Max((Float)0, 1 - cosTheta * cosTheta));
        Float phi = 2 * ORC_Pi * u.y;
        V3 wh(sinTheta * std::cos(phi), sinTheta * std::sin(phi), cosTheta); /* SphericalDirection, geometry.h:1470 */
        if (!SameHemisphere(wo, wh)) wh = -wh;
        *wi = Reflect(wo, wh);
        if (!SameHemisphere(wo, *wi)) return V3(0.f);
        *pdf = BxDF_Pdf(b, wo, *wi);
        return BxDF_f(b, wo, *wi);
    }
    case LambertianTransmission:
        *wi = CosineSampleHemisphere(u);
        if (wo.z > 0) wi->z *= -1;
        *pdf = BxDF_Pdf(b, wo, *wi);
        return BxDF_f(b, wo, *wi);
    case SpecularReflection:
        *wi = V3(-wo.x, -wo.y, wo.z);
        *pdf = 1;
        return FresnelEvaluate(b.fresnel, CosTheta(*wi)) * b.S / AbsCosTheta(*wi);
    case SpecularTransmission: {
        bool entering = CosTheta(wo) > 0;
        Float etaI = entering ? b.A : b.B;
        Float etaT = entering ? b.B : b.A;
        if (!Refract(wo, Faceforward(V3(0, 0, 1), wo), etaI / etaT, wi)) return V3(0);
        *pdf = 1;
        V3 ft = b.S * (V3(1.) - FresnelEvaluate(b.fresnel, CosTheta(*wi)));
        ft *= (etaI * etaI) / (etaT * etaT);
        return ft / AbsCosTheta(*wi);
    }
    case FresnelSpecular: {
        Float F = FrDieletric(CosTheta(wo), b.A, b.B);
        if (u[0] < F) {
            *wi = V3(-wo.x, -wo.y, wo.z);
            if (sampledType) *sampledType = BSDF_SPECULAR | BSDF_REFLECTION;
            *pdf = F;
            return F * b.S / AbsCosTheta(*wi);
        } else {
            bool entering = CosTheta(wo) > 0;
            Float etaI = entering ? b.A : b.B;
            Float etaT = entering ? b.B : b.A;
            if (!Refract(wo, Faceforward(V3(0, 0, 1), wo), etaI / etaT, wi)) return V3(0);
            V3 ft = b.T * (1 - F);
            ft *= (etaI * etaI) / (etaT * etaT);
            if (sampledType) *sampledType = BSDF_SPECULAR | BSDF_TRANSMISSION;
            *pdf = 1 - F;
            return ft / AbsCosTheta(*wi);
        }
    }
    case MicrofacetTransmission: {
        if (IsZero(wo.z)) return V3(0.);
        V3 wh = MF_Sample_wh(b.mDist, wo, u);
        if (Dot(wo, wh) < 0) return V3(0.);
        Float eta = CosTheta(wo) > 0 ? (b.A / b.B) : (b.B / b.A);
        if (!Refract(wo, wh, eta, wi)) return V3(0);
        *pdf = BxDF_Pdf(b, wo, *wi);
        return BxDF_f(b, wo, *wi);
    }
    case MicrofacetReflection: {
        if (IsZero(wo.z)) return V3(0.);
        V3 wh = MF_Sample_wh(b.mDist, wo, u);
        if (Dot(wo, wh) < 0) return V3(0.);
        *wi = Reflect(wo, wh);
        if (!SameHemisphere(wo, *wi)) return V3(0.f);
        Float dd = Dot(wo, wh);
        *pdf = MF_Pdf(b.mDist, wo, wh) / (4 * dd);
        return BxDF_f(b, wo, *wi);
    }
    case FresnelBlend: {
        V2 uu = u;
        if (uu[0] < .5) {
            uu[0] = Min(2 * uu[0], ORC_OneMinusEpsilon);
            *wi = CosineSampleHemisphere(uu);
            if (wo.z < 0) wi->z *= -1;
        } else {
            uu[0] = Min(2 * (uu[0] - .5f), ORC_OneMinusEpsilon);
            V3 wh = MF_Sample_wh(b.mDist, wo, uu);
            *wi = Reflect(wo, wh);
            if (!SameHemisphere(wo, *wi)) return V3(0.f);
        }
        *pdf = BxDF_Pdf(b, wo, *wi);
        return BxDF_f(b, wo, *wi);
    }
    default: *pdf = 0; return V3(0.f);
    }
}

/* ---- reflection.h:367-372, reflection.cpp:591-701 ---------------------------------------------- */
BSDF::BSDF(const SurfaceInteraction &si) : has_bssrdf(0), nBxDFs(0) {
    ns = si.n;
    ng = si.n;
    ss = Normalize(si.dpdu);
    ts = Cross(ns, ss);
}
int BSDF::NumComponents(int flags) const {
    int num = 0;
    for (int i = 0; i < nBxDFs; ++i) if (bxdfs[i].MatchesFlags(flags)) ++num;
    return num;
}
V3 BSDF::Sample_f(const V3 &woWorld, V3 *wiWorld, const V2 &u, Float *pdf, int type, int *sampledType) const {
    int matchingComps = NumComponents(type);
    if (matchingComps == 0) {
        *pdf = 0;
        if (sampledType) *sampledType = 0;
        return V3(0);
    }
    int comp = (int)Min((Float)(int)std::floor(u[0] * matchingComps), (Float)(matchingComps - 1)); /* Min(Float,Float) */
    const BxDF *bxdf = nullptr;
    int count = comp, picked = -1;
    for (int i = 0; i < nBxDFs; ++i) {
        if (bxdfs[i].MatchesFlags(type) && count-- == 0) { bxdf = &bxdfs[i]; picked = i; break; }
    }
    V2 uRemapped(Min(u[0] * matchingComps - comp, ORC_OneMinusEpsilon), u[1]);
    V3 wi, wo = WorldToLocal(woWorld);
    if (IsZero(wo.z)) return V3(0.);
    *pdf = 0;
    if (sampledType) *sampledType = bxdf->type;
    V3 f = BxDF_Sample_f(*bxdf, wo, &wi, uRemapped, pdf, sampledType);
    if (IsZero(*pdf)) {
        if (sampledType) *sampledType = 0;
        return V3(0);
    }
    *wiWorld = LocalToWorld(wi);
    if (!(bxdf->type & BSDF_SPECULAR) && matchingComps > 1) {
        for (int i = 0; i < nBxDFs; ++i)
            if (i != picked && bxdfs[i].MatchesFlags(type)) *pdf += BxDF_Pdf(bxdfs[i], wo, wi);
    }
    if (matchingComps > 1) *pdf /= matchingComps;
    if (!(bxdf->type & BSDF_SPECULAR)) {
        bool reflect = Dot(*wiWorld, ng) * Dot(woWorld, ng) > 0;
        f = V3(0.);
        for (int i = 0; i < nBxDFs; ++i) {
            if (bxdfs[i].MatchesFlags(type) &&
                ((reflect && (bxdfs[i].type & BSDF_REFLECTION)) || (!reflect && (bxdfs[i].type & BSDF_TRANSMISSION))))
                f += BxDF_f(bxdfs[i], wo, wi);
        }
    }
    return f;
}
V3 BSDF::f(const V3 &woW, const V3 &wiW, int flags) const {
    if (nBxDFs == 0) return V3(0);
    V3 wi = WorldToLocal(wiW), wo = WorldToLocal(woW);
    if (IsZero(wo.z)) return V3(0.);
    bool reflect = Dot(wiW, ng) * Dot(woW, ng) > 0;
    V3 f(0.f);
    for (int i = 0; i < nBxDFs; ++i) {
        if (bxdfs[i].MatchesFlags(flags) &&
            ((reflect && (bxdfs[i].type & BSDF_REFLECTION)) || (!reflect && (bxdfs[i].type & BSDF_TRANSMISSION))))
            f += BxDF_f(bxdfs[i], wo, wi);
    }
    return f;
}
Float BSDF::Pdf(const V3 &woWorld, const V3 &wiWorld, int flags) const {
    if (nBxDFs == 0.f) return 0.f;
    V3 wo = WorldToLocal(woWorld), wi = WorldToLocal(wiWorld);
    if (IsZero(wo.z)) return 0.;
    Float pdf = 0.f;
    int matchingComps = 0;
    for (int i = 0; i < nBxDFs; i++) {
        if (bxdfs[i].MatchesFlags(flags)) { pdf += BxDF_Pdf(bxdfs[i], wo, wi); matchingComps += 1; }
    }
    Float v = matchingComps > 0 ? pdf / matchingComps : 0.f;
    return v;
}

/* ---- texture.h:18-84 ------------------------------------------------------------------------------ */
static Float InverseGammaCorrect(Float value) { /* geometry.h:79-82 */
    if (value <= 0.04045f) return value * 1.f / 12.92f;
    return powf((value + 0.055f) * 1.f / 1.055f, (Float)2.4f);
}
static V3 ImageLookup(const Scene &sc, const Tex &t, const SurfaceInteraction *si) {
    const ImageTex &im = sc.images[t.image];
    int i = si->uv.x * im.width;
    int j = si->uv.y * im.height;
    if (i < 0) i = 0;
    if (i > im.width - 1) i = im.width - 1;
    if (j < 0) j = 0;
    if (j > im.height - 1) j = im.height - 1;
    V3 rgb = im.data[i + im.width * j];
    V3 out;
    for (int c = 0; c < 3; ++c) out[c] = 1.f * InverseGammaCorrect(rgb[c]);
    return out;
}
/* SpectrumTexture / FloatTexture of scene.cpp:769-788 */
static V3 EvalSpec(const Scene &sc, const Tex &t, const SurfaceInteraction *si) {
    if (t.type == 0) return t.spec;
    if (t.type == 1) return V3(t.fval);
    return ImageLookup(sc, t, si);
}
static Float EvalFloat(const Scene &sc, const Tex &t, const SurfaceInteraction *si) {
    if (t.type == 0) return t.spec[0];
    if (t.type == 1) return t.fval;
    return ImageLookup(sc, t, si)[0];
}

/* ---- BxDF initialisers, reflection.h:131-246 ------------------------------------------------------- */
static BxDF MkLambertR(const V3 &R) { BxDF b; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.impl = LambertianReflection; b.S = R; return b; }
static BxDF MkLambertT(const V3 &T) { BxDF b; b.type = BSDF_TRANSMISSION | BSDF_DIFFUSE; b.impl = LambertianTransmission; b.S = T; return b; }
static BxDF MkOrenNayar(const V3 &R, Float sigma) {
    BxDF b; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.impl = OrenNayar; b.S = R;
    Float sigma2 = sigma * sigma;
    b.A = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
    b.B = 0.45f * sigma2 / (sigma2 + 0.09f);
    return b;
}
static Fresnel MkDielectric(Float etaI, Float etaT) { Fresnel f; f.type = FrDielectric; f.etaI = etaI; f.etaT = etaT; return f; }
static BxDF MkSpecR(const V3 &R, const Fresnel &fr) { BxDF b; b.type = BSDF_REFLECTION | BSDF_SPECULAR; b.impl = SpecularReflection; b.S = R; b.fresnel = fr; return b; }
static BxDF MkSpecT(const V3 &T, Float etaA, Float etaB) {
    BxDF b; b.type = BSDF_TRANSMISSION | BSDF_SPECULAR; b.impl = SpecularTransmission; b.S = T; b.A = etaA; b.B = etaB;
    b.fresnel = MkDielectric(etaA, etaB);
    return b;
}
static BxDF MkFresnelSpec(const V3 &R, const V3 &T, Float eA, Float eB) {
    BxDF b; b.type = BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR; b.impl = FresnelSpecular; b.S = R; b.T = T; b.A = eA; b.B = eB;
    return b;
}
static MicrofacetDistribution MkDist(Float ax, Float ay, bool remap) {
    MicrofacetDistribution m;
    m.alphax = remap ? RoughnessToAlpha(ax) : ax;
    m.alphay = remap ? RoughnessToAlpha(ay) : ay;
    m.sampleVisibleArea = true;
    return m;
}
static BxDF MkMicroT(const V3 &T, Float eA, Float eB, Float ax, Float ay, bool remap = true) {
    BxDF b; b.type = BSDF_TRANSMISSION | BSDF_GLOSSY; b.impl = MicrofacetTransmission; b.A = eA; b.B = eB; b.T = T;
    b.fresnel = MkDielectric(eA, eB);
    b.mDist = MkDist(ax, ay, remap);
    return b;
}
static BxDF MkMicroR(const V3 &R, Float ax, Float ay, const Fresnel &fr, bool remap = true) {
    BxDF b; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.impl = MicrofacetReflection; b.S = R; b.fresnel = fr;
    b.mDist = MkDist(ax, ay, remap);
    return b;
}

/* glass.cpp:14-58 and the identical glossy part of subsurface.cpp:27-67 / kdsubsurface.cpp:118-158
 * (those pass remap=false and push R before T) */
static void GlassLobes(BSDF *bsdf, const V3 &R, const V3 &T, Float eta, Float urough, Float vrough, bool subsurfaceOrder) {
    bool isSpecular = IsZero(urough) && IsZero(vrough);
    if (isSpecular) { /* mLobes is always true on this path */
        bsdf->Push(MkFresnelSpec(R, T, 1.f, eta));
        return;
    }
    bool remap = !subsurfaceOrder;
    if (subsurfaceOrder) {
        if (!IsBlack(R)) bsdf->Push(MkMicroR(R, urough, vrough, MkDielectric(1.f, eta), remap));
        if (!IsBlack(T)) bsdf->Push(MkMicroT(T, 1.f, eta, urough, vrough, remap));
    } else {
        if (!IsBlack(T)) bsdf->Push(MkMicroT(T, 1.f, eta, urough, vrough, remap));
        if (!IsBlack(R)) bsdf->Push(MkMicroR(R, urough, vrough, MkDielectric(1.f, eta), remap));
    }
}
static V3 Clamp01(const V3 &v) { /* geometry.h:582-585: Clamp(val) clamps to [-1,1] */
    return V3(Clampf(v.x, -1, 1), Clampf(v.y, -1, 1), Clampf(v.z, -1, 1));
}

void ComputeScatteringFunctions(const Scene &sc, SurfaceInteraction *si, BSDF *bsdf) {
    const Primitive &pr = sc.prims[si->primitive];
    if (pr.emissive) { /* primitive.cpp:89-96: one invalidated lobe whose type is never initialised.
                          Restated as type 0 + is_valid 0 — see DESIGN.md "emitter lobe". */
        BxDF b; b.is_valid = 0; b.type = 0x40000000; /* matches no flag set, as garbage would */
        bsdf->Push(b);
        return;
    }
    if (pr.material < 0) return; /* primitive.cpp:98-106 */
    const Material &m = sc.materials[pr.material];
    const Tex *tx = m.tex;
    switch (m.type) {
    case Matte: { /* matte.cpp:10-27 */
        V3 kd = EvalSpec(sc, tx[0], si);
        Float sig = EvalFloat(sc, tx[1], si);
        if (!IsBlack(kd)) {
            if (IsZero(sig)) bsdf->Push(MkLambertR(kd));
            else bsdf->Push(MkOrenNayar(kd, sig));
        }
    } break;
    case Mirror: { /* mirror.cpp:9-20 */
        V3 kr = EvalSpec(sc, tx[0], si);
        if (!IsBlack(kr)) bsdf->Push(MkSpecR(kr, Fresnel()));
    } break;
    case Glass: { /* glass.cpp:14-58; slots scene.cpp:814-820 */
        Float eta = EvalFloat(sc, tx[4], si);
        Float urough = EvalFloat(sc, tx[2], si), vrough = EvalFloat(sc, tx[3], si);
        V3 R = EvalSpec(sc, tx[0], si), T = EvalSpec(sc, tx[1], si);
        if (IsBlack(R) && IsBlack(T)) return;
        GlassLobes(bsdf, R, T, eta, urough, vrough, false);
    } break;
    case Metal: { /* metal.cpp:12-25 */
        Float urough = EvalFloat(sc, tx[2], si), vrough = EvalFloat(sc, tx[3], si);
        Fresnel fr; fr.type = FrConductorT; fr.setaI = V3(1.f); fr.setaT = EvalSpec(sc, tx[0], si); fr.sk = EvalSpec(sc, tx[1], si);
        bsdf->Push(MkMicroR(V3(1.f), urough, vrough, fr));
    } break;
    case Plastic: { /* plastic.cpp:12-32 */
        V3 kd = EvalSpec(sc, tx[0], si);
        if (!IsBlack(kd)) bsdf->Push(MkLambertR(kd));
        V3 ks = EvalSpec(sc, tx[1], si);
        if (!IsBlack(ks)) {
            Float rough = EvalFloat(sc, tx[2], si);
            bsdf->Push(MkMicroR(ks, rough, rough, MkDielectric(1.5f, 1.f)));
        }
    } break;
    case Uber: { /* uber.cpp:18-65; slots scene.cpp:836-845 */
        Float e = EvalFloat(sc, tx[7], si);
        V3 op = EvalSpec(sc, tx[6], si);
        V3 t = (-op + V3(1.f));
        if (!IsBlack(t)) bsdf->Push(MkSpecT(t, 1.f, 1.f));
        V3 kd = op * EvalSpec(sc, tx[0], si);
        if (!IsBlack(kd)) bsdf->Push(MkLambertR(kd));
        V3 ks = op * EvalSpec(sc, tx[1], si);
        if (!IsBlack(ks)) {
            Float roughu = EvalFloat(sc, tx[4], si), roughv = EvalFloat(sc, tx[5], si);
            bsdf->Push(MkMicroR(ks, roughu, roughv, MkDielectric(1.f, e)));
        }
        V3 kr = op * EvalSpec(sc, tx[2], si);
        if (!IsBlack(kr)) bsdf->Push(MkSpecR(kr, MkDielectric(1.f, e)));
        V3 kt = op * EvalSpec(sc, tx[3], si);
        if (!IsBlack(kt)) bsdf->Push(MkSpecT(kt, 1.f, e));
    } break;
    case Translucent: { /* translucent.cpp:20-64 */
        Float eta = 1.5f;
        V3 r = EvalSpec(sc, tx[2], si), t = EvalSpec(sc, tx[3], si);
        if (IsBlack(r) && IsBlack(t)) return;
        V3 kd = EvalSpec(sc, tx[0], si);
        if (!IsBlack(kd)) {
            if (!IsBlack(r)) bsdf->Push(MkLambertR(r * kd));
            if (!IsBlack(t)) bsdf->Push(MkLambertT(t * kd));
        }
        V3 ks = EvalSpec(sc, tx[1], si);
        if (!IsBlack(ks) && (!IsBlack(r) || !IsBlack(t))) {
            Float rough = EvalFloat(sc, tx[4], si);
            if (!IsBlack(r)) bsdf->Push(MkMicroR(r * ks, rough, rough, MkDielectric(1.f, eta)));
            if (!IsBlack(t)) bsdf->Push(MkMicroT(t * ks, 1.f, eta, rough, rough));
        }
    } break;
    case Subsurface: { /* subsurface.cpp:21-74; ctor args scene.cpp:847-856: g=vals[2], eta=vals[1], scale=vals[0] */
        Float eta = m.vals[1], scale = m.vals[0];
        V3 R = Clamp01(EvalSpec(sc, tx[0], si)), T = Clamp01(EvalSpec(sc, tx[1], si));
        Float urough = EvalFloat(sc, tx[4], si), vrough = EvalFloat(sc, tx[5], si);
        if (IsBlack(R) && IsBlack(T)) return;
        GlassLobes(bsdf, R, T, eta, urough, vrough, true);
        V3 sig_a = scale * EvalSpec(sc, tx[2], si);
        V3 sig_s = scale * EvalSpec(sc, tx[3], si);
        InitTabulatedBSSRDF(&bsdf->bssrdf, *si, eta, pr.material, sig_a, sig_s, &sc.tables[m.bssrdfTable]);
        bsdf->has_bssrdf = 1;
    } break;
    case KdSubsurface: { /* kdsubsurface.cpp:39-94; ctor args scene.cpp:858-866: (kd,kr,kt,mfp,...) = tex[0..3], e=vals[0], scale=vals[1] */
        Float eta = m.vals[0], scale = m.vals[1];
        V3 R = Clamp01(EvalSpec(sc, tx[1], si)), T = Clamp01(EvalSpec(sc, tx[2], si));
        Float urough = EvalFloat(sc, tx[4], si), vrough = EvalFloat(sc, tx[5], si);
        if (IsBlack(R) && IsBlack(T)) return;
        GlassLobes(bsdf, R, T, eta, urough, vrough, true);
        V3 kd = EvalSpec(sc, tx[0], si);
        V3 mfree = scale * EvalSpec(sc, tx[3], si);
        V3 sig_a, sig_s;
        SubsurfaceFromDiffuse(sc.tables[m.bssrdfTable], kd, mfree, &sig_a, &sig_s);
        InitTabulatedBSSRDF(&bsdf->bssrdf, *si, eta, pr.material, sig_a, sig_s, &sc.tables[m.bssrdfTable]);
        bsdf->has_bssrdf = 1;
    } break;
    case Disney: { /* No material of the reference pushes its Disney lobes (reflection.h:90-129 are never called); this is
                      how PBRT-v3's DisneyMaterial assembles the same five, without the metallic / transmission / specular
                      terms the reference has no lobes for.  Slots: 0 colour, 1 roughness, 2 sheen, 3 sheenTint, 4 clearcoat,
                      5 clearcoatGloss, 6 flatness.  Init_DisneyClearcoat stores its arguments in clearCoat / alpha while
                      f / Pdf / Sample_f read A / B (reflection.h:90-95 vs reflection.cpp:82,291): restated with the
                      two numbers where they are read. */
        V3 c = EvalSpec(sc, tx[0], si);
        Float rough = EvalFloat(sc, tx[1], si), sheen = EvalFloat(sc, tx[2], si), stint = EvalFloat(sc, tx[3], si);
        Float cc = EvalFloat(sc, tx[4], si), gloss = EvalFloat(sc, tx[5], si), flat = EvalFloat(sc, tx[6], si);
        if (!IsBlack(c)) {
            BxDF b;
            b.type = BSDF_REFLECTION | BSDF_DIFFUSE;
            b.impl = DisneyDiffuse; b.S = (1 - flat) * c; bsdf->Push(b);
            if (flat > 0) { b.impl = DisneyFakeSS; b.S = flat * c; b.A = rough; bsdf->Push(b); }
            b.impl = DisneyRetro; b.S = c; b.A = rough; bsdf->Push(b);
            if (sheen > 0) { b.impl = DisneySheen; b.S = c; b.sheen = V3(sheen); b.sheenTint = stint; bsdf->Push(b); }
        }
        if (cc > 0) {
            BxDF b;
            b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.impl = DisneyClearcoat;
            b.A = cc; b.B = (1 - gloss) * .1f + gloss * .001f;
            bsdf->Push(b);
        }
    } break;
    case PlayGround: { /* playground.cpp:6-23 */
        V3 base = EvalSpec(sc, tx[0], si);
        for (int i = 0; i < m.ntex; i++) {
            if (si->faceIndex == i) { base = EvalSpec(sc, tx[i], si); break; }
        }
        bsdf->Push(MkLambertR(base));
    } break;
    default: break;
    }
}

/* ---- unit hooks (tests only) ---- */
Float Unit_FrDielectric(Float cosI, Float etaI, Float etaT) { return FrDieletric(cosI, etaI, etaT); }
V3 Unit_FrConductor(Float cosI, const V3 &etaI, const V3 &etaT, const V3 &k) { return FrConductor(cosI, etaI, etaT, k); }
static MicrofacetDistribution unit_dist(Float ax, Float ay) { MicrofacetDistribution m; m.alphax = ax; m.alphay = ay; m.sampleVisibleArea = true; return m; }
Float Unit_GGX_D(Float ax, Float ay, const V3 &wh) { return MF_D(unit_dist(ax, ay), wh); }
Float Unit_GGX_G1(Float ax, Float ay, const V3 &w) { return MF_G1(unit_dist(ax, ay), w); }
Float Unit_GGX_Pdf(Float ax, Float ay, const V3 &wo, const V3 &wh) { return MF_Pdf(unit_dist(ax, ay), wo, wh); }
V3 Unit_GGX_Sample_wh(Float ax, Float ay, const V3 &wo, const V2 &u) { return MF_Sample_wh(unit_dist(ax, ay), wo, u); }

} // namespace orc
