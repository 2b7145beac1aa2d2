/*
 * ptr_tables.cpp — host-side build of the subsurface scattering profile table a Subsurface /
 * KdSubsurface material carries (the reference fills a BSSRDFTable of 100 albedos x 64 radii with
 * photon beam diffusion when the material is made: bssrdf.cpp:193-298, called from
 * subsurface.cpp:15-16 and kdsubsurface.cpp:33).  Single precision with the reference's promotion
 * points (Pi and Inv4Pi are double literals), so the table is the same set of floats.
 */
#include "ptr_host.h"
#include <cmath>

namespace {

const double PI_D = 3.14159265358979323846;
const double INV_4PI_D = 0.07957747154594766788;
const int BEAM_SAMPLES = 100;

float moment1(float e) { /* bssrdf.cpp:7-16 */
    float e2 = e * e, e3 = e2 * e, e4 = e3 * e, e5 = e4 * e;
    if (e < 1) return 0.45966f - 1.73965f * e + 3.37668f * e2 - 3.904945 * e3 + 2.49277f * e4 - 0.68441f * e5;
    return -4.61686f + 11.1136f * e - 10.4646f * e2 + 5.11455f * e3 - 1.27198f * e4 + 0.12746f * e5;
}
float moment2(float e) { /* bssrdf.cpp:18-30 */
    float e2 = e * e, e3 = e2 * e, e4 = e3 * e, e5 = e4 * e;
    if (e < 1) return 0.27614f - 0.87350f * e + 1.12077f * e2 - 0.65095f * e3 + 0.07883f * e4 + 0.04860f * e5;
    float r = 1 / e, r2 = r * r, r3 = r2 * r;
    return -547.033f + 45.3087f * r3 - 218.725f * r2 + 458.843f * r + 404.557f * e - 189.519f * e2 + 54.9327f * e3 - 9.00603f * e4 + 0.63942f * e5;
}
float dielectric(float ci, float ni, float nt) { /* material.cpp:5-28 */
    ci = ci < -1 ? -1.f : (ci > 1 ? 1.f : ci);
    if (!(ci > 0.f)) {
        float s = ni; ni = nt; nt = s;
        ci = ci > 0 ? ci : -ci;
    }
    float a = 1 - ci * ci;
    float si = sqrtf(a > 0 ? a : 0.f);
    float st = ni / nt * si;
    if (st >= 1) return 1;
    float b = 1 - st * st;
    float ct = sqrtf(b > 0 ? b : 0.f);
    float par = ((nt * ci) - (ni * ct)) / ((nt * ci) + (ni * ct));
    float per = ((ni * ci) - (nt * ct)) / ((ni * ci) + (nt * ct));
    return (par * par + per * per) / 2;
}
float hg(float c, float g) { /* medium.cpp:4-19 */
    float den = 1 + g * g + 2 * g * c;
    return INV_4PI_D * (1 - g * g) / (den * sqrtf(den));
}

/* multiple scattering term, bssrdf.cpp:193-236 */
float beam_ms(float ss, float sa, float g, float eta, float r) {
    float acc = 0;
    float sps = ss * (1 - g);
    float spt = sa + sps;
    float albedo = sps / spt;
    float Dg = (2 * sa + sps) / (3 * spt * spt);
    float str = sqrtf(sa / Dg);
    float m1 = moment1(eta), m2 = moment2(eta);
    float ze = -2 * Dg * (1 + 3 * m2) / (1 - 2 * m1);
    float cPhi = .25f * (1 - 2 * m1), cE = .5f * (1 - 3 * m2);
    for (int i = 0; i < BEAM_SAMPLES; ++i) {
        float zr = -logf(1 - (i + .5f) / BEAM_SAMPLES) / spt;
        float zv = -zr + 2 * ze;
        float dr = sqrtf(r * r + zr * zr), dv = sqrtf(r * r + zv * zv);
        float phiD = INV_4PI_D / Dg * (expf(-str * dr) / dr - expf(-str * dv) / dv);
        float EDn = INV_4PI_D * (zr * (1 + str * dr) * expf(-str * dr) / (dr * dr * dr) - zv * (1 + str * dv) * expf(-str * dv) / (dv * dv * dv));
        float E = phiD * cPhi + EDn * cE;
        float kappa = 1 - expf(-2 * spt * (dr + zr));
        acc += kappa * albedo * albedo * E;
    }
    return acc / BEAM_SAMPLES;
}
/* single scattering term, bssrdf.cpp:238-262 */
float beam_ss(float ss, float sa, float g, float eta, float r) {
    float st = sa + ss, albedo = ss / st;
    float tc = r * sqrtf(eta * eta - 1);
    float acc = 0;
    for (int i = 0; i < BEAM_SAMPLES; ++i) {
        float ti = tc - logf(1 - (i + .5f) / BEAM_SAMPLES) / st;
        float d = sqrtf(r * r + ti * ti);
        float co = ti / d;
        acc += albedo * expf(-st * (d + tc)) / (d * d) * hg(co, g) * (1 - dielectric(-co, 1, eta)) * std::abs(co);
    }
    return acc / BEAM_SAMPLES;
}
/* bssrdf.cpp:300-326: integral of the Catmull-Rom spline through (x, f), running sums in cdf */
float spline_integral(int n, const float *x, const float *f, float *cdf) {
    float sum = 0;
    cdf[0] = 0;
    for (int i = 0; i + 1 < n; ++i) {
        float w = x[i + 1] - x[i];
        float d0 = i > 0 ? w * (f[i + 1] - f[i - 1]) / (x[i + 1] - x[i - 1]) : f[i + 1] - f[i];
        float d1 = i + 2 < n ? w * (f[i + 2] - f[i]) / (x[i + 2] - x[i]) : f[i + 1] - f[i];
        sum += ((d0 - d1) * (1.f / 12.f) + (f[i] + f[i + 1]) * .5f) * w;
        cdf[i + 1] = sum;
    }
    return sum;
}

} // namespace

/* ComputeBeamDiffusionBSSRDF, bssrdf.cpp:264-298 */
void ptr_build_bssrdf_table(float g, float eta, HostTable *t) {
    const int nr = t->n_rho = 100, nq = t->n_radius = 64;
    t->rho.assign(nr, 0.f);
    t->radius.assign(nq, 0.f);
    t->profile.assign((size_t)nr * nq, 0.f);
    t->rho_eff.assign(nr, 0.f);
    t->profile_cdf.assign((size_t)nr * nq, 0.f);
    t->radius[1] = 2.5e-3f;
    for (int j = 2; j < nq; ++j) t->radius[j] = t->radius[j - 1] * 1.2f;
    for (int i = 0; i < nr; ++i) t->rho[i] = (1 - expf(-8 * i / (float)(nr - 1))) / (1 - expf(-8.f));
    for (int i = 0; i < nr; i++) {
        float *row = &t->profile[(size_t)i * nq];
        for (int j = 0; j < nq; ++j) {
            float rho = t->rho[i], r = t->radius[j];
            row[j] = 2 * PI_D * r * (beam_ss(rho, 1 - rho, g, eta, r) + beam_ms(rho, 1 - rho, g, eta, r));
        }
        t->rho_eff[i] = spline_integral(nq, t->radius.data(), row, &t->profile_cdf[(size_t)i * nq]);
    }
}
