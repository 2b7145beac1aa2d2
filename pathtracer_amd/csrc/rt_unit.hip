/*
 * rt_unit.hip — unit hooks of the device code: the building blocks of the shade stage evaluated on plain arrays, so
 * that the known-answer tests (tests/test_gpu_pins.py: normalisation integrals, closed forms, sample histograms) can
 * hold the HIP path to the same answers as the CPU checker.  Kinds and record layouts: include/ptr_api.h (PTR_UNIT_*).
 * Not on the render path.
 */
#include "rt_common.h"
#include "dev_bssrdf.h"

PD SI unit_si(int prim, float u, float v) { /* a canonical interaction: n = +z, dpdu = +x */
    SI si;
    si.p = f3(0, 0, 0); si.pError = f3(0, 0, 0); si.wo = f3(0, 0, 1);
    si.n = f3(0, 0, 1);
    si.dpdu = f3(1, 0, 0);
    si.uv = f2(u, v);
    si.prim = prim; si.face = 0;
    si.med_in = -1; si.med_out = -1;
    return si;
}

__global__ void __launch_bounds__(TPB) k_unit(DScene sc, int prim, int kind, int n, const float *in, int in_w, float *out, int out_w) {
    __shared__ float s_lobes[LOBE_LDS_WORDS];
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const float *a = in + (size_t)i * in_w;
    float *o = out + (size_t)i * out_w;
    switch (kind) {
    case 0: o[0] = fr_dielectric(a[0], a[1], a[2]); break;
    case 1: { f3 r = fr_conductor(a[0], f3(a[1], a[2], a[3]), f3(a[4], a[5], a[6]), f3(a[7], a[8], a[9])); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
    case 2: o[0] = mf_D(a[0], a[1], f3(a[2], a[3], a[4])); break;
    case 3: o[0] = mf_G1(a[0], a[1], f3(a[2], a[3], a[4])); break;
    case 4: o[0] = mf_pdf(a[0], a[1], f3(a[2], a[3], a[4]), f3(a[5], a[6], a[7])); break;
    case 5: { f3 r = mf_sample_wh(a[0], a[1], f3(a[2], a[3], a[4]), f2(a[5], a[6])); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
    case 6: o[0] = phase_hg(a[0], a[1]); break;
    case 7: { f3 wi; float p = phase_sample_p(a[0], f3(a[1], a[2], a[3]), &wi, f2(a[4], a[5])); o[0] = wi.x; o[1] = wi.y; o[2] = wi.z; o[3] = p; } break;
    case 8: o[0] = roughness_to_alpha(a[0]); break;
    case 9: { float pdf = 0; f2 uv = d2_sample(sc, f2(a[0], a[1]), &pdf); o[0] = uv.x; o[1] = uv.y; o[2] = pdf; } break;
    case 10: o[0] = d2_pdf(sc, f2(a[0], a[1])); break;
    case 11: case 12: case 13: {
        SI si = unit_si(prim, 0.5f, 0.5f);
        BSDF bsdf;
        bsdf.lm = &s_lobes[threadIdx.x];
        bsdf_init(&bsdf, si);
        BssrdfParams bp;
        compute_scattering(sc, si, &bsdf, &bp);
        Bssrdf b;
        bssrdf_make(sc, bp, si, bsdf, &b);
        if (kind == 11) { f3 r = bssrdf_sr(b, a[0]); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
        else if (kind == 12) o[0] = bssrdf_pdf_sr(b, (int)a[0], a[1]);
        else o[0] = bssrdf_sample_sr(b, (int)a[0], a[1]);
    } break;
    default: break;
    }
}

/* BSDF::f / Pdf over the non-specular lobes (sample = 0), or BSDF::Sample_f over all lobes (sample = 1), of primitive
 * `prim`'s material at the canonical interaction; directions in world space */
__global__ void __launch_bounds__(TPB) k_unit_bsdf(DScene sc, int prim, int sample, int n, const float *wo, const float *wi_or_u, const float *uv,
                                                   float *wi_out, float *f, float *pdf, int *flags) {
    __shared__ float s_lobes[LOBE_LDS_WORDS];
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    SI si = unit_si(prim, uv[2 * i], uv[2 * i + 1]);
    BSDF bsdf;
    bsdf.lm = &s_lobes[threadIdx.x];
    bsdf_init(&bsdf, si);
    BssrdfParams bp;
    compute_scattering(sc, si, &bsdf, &bp);
    const f3 o(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]);
    if (!sample) {
        const f3 w(wi_or_u[3 * i], wi_or_u[3 * i + 1], wi_or_u[3 * i + 2]);
        const f3 r = bsdf_f(bsdf, o, w, BSDF_ALL & ~BSDF_SPECULAR);
        f[3 * i] = r.x; f[3 * i + 1] = r.y; f[3 * i + 2] = r.z;
        pdf[i] = bsdf_pdf(bsdf, o, w, BSDF_ALL & ~BSDF_SPECULAR);
    } else {
        f3 w(0, 0, 0);
        float p = 0;
        int fl = 0;
        const f3 r = bsdf_sample_f(bsdf, o, &w, f2(wi_or_u[2 * i], wi_or_u[2 * i + 1]), &p, BSDF_ALL, &fl);
        wi_out[3 * i] = w.x; wi_out[3 * i + 1] = w.y; wi_out[3 * i + 2] = w.z;
        f[3 * i] = r.x; f[3 * i + 1] = r.y; f[3 * i + 2] = r.z;
        pdf[i] = p;
        flags[i] = fl;
    }
}

void rt_launch_unit(hipStream_t s, const DScene &sc, int prim, int kind, int n, const float *in, int in_w, float *out, int out_w) {
    hipLaunchKernelGGL(k_unit, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, s, sc, prim, kind, n, in, in_w, out, out_w);
}
void rt_launch_unit_bsdf(hipStream_t s, const DScene &sc, int prim, int sample, int n, const float *wo, const float *wi_or_u, const float *uv, float *wi_out,
                         float *f, float *pdf, int *flags) {
    hipLaunchKernelGGL(k_unit_bsdf, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, s, sc, prim, sample, n, wo, wi_or_u, uv, wi_out, f, pdf, flags);
}
