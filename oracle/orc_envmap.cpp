/*
 * orc_envmap.cpp — ORACLE (test infrastructure, never linked into the product): CPU restatement of the
 * reference's light-map bring-up, i.e. what InsertEXRLightMap does to the pixels of an .exr before the
 * render path samples them:
 *     BuildSpectrumMipMap   src/core/texture.cpp:217-260   scale, MipMap, sin-weighted luminance table
 *     BuildMipMap           src/core/texture.cpp:111-215   Lanczos resample to powers of two, box pyramid
 *     GetResampleWeights    src/core/texture.cpp:91-109,   Lanczos :19-27
 *     Texel / MipMap::triangle / MipMap::Lookup   texture.cpp:68-88, texture.h:132-179
 *     CreateDistribution1D / 2D                   src/lights/infinite.cpp:7-39
 * Both product builders - the host mirror's InsertLightMap (pathtracer_amd/host/pt_scene.cpp) and the device
 * pipeline ptr_envmap_build (csrc/rt_envmap.hip) - are compared with these tables bit for bit
 * (tests/test_cpu_envmap_oracle.py, tests/test_gpu_envmap_build.py).
 *
 * PARITY UNPINNED by the reference (it holds no tests, no .exr and no tables); pinned here by closed forms:
 * a constant map gives func = Y * sin(theta_v), row CDFs that are straight lines and a marginal CDF that
 * follows (1 - cos theta) / 2 (tests/test_cpu_envmap_oracle.py).
 *
 * Literal points kept: `Pi` is a double literal (geometry.h:35), so `x *= Pi` and `Pi * (v + .5f) / height`
 * are evaluated in double and narrowed where the reference narrows (assignment to Float, the Float parameter
 * of Sin, geometry.h:658); Log2 is log(x) * invLog2 in float (geometry.h:94-97); the resample path clamps
 * texels to [0, 1] (texture.cpp:166) while the power-of-two path copies them unclamped (:174-177).
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

typedef float Float;
static const double kPiD = 3.14159265358979323846; /* geometry.h:35 */

struct Rgb {
    Float r, g, b;
};
static inline Rgb operator+(const Rgb &a, const Rgb &b) { return Rgb{a.r + b.r, a.g + b.g, a.b + b.b}; }
static inline Rgb operator*(Float s, const Rgb &a) { return Rgb{s * a.r, s * a.g, s * a.b}; }

/* texture.cpp:19-27 with tau = 2 (texture.h:94) */
static Float lanczos(Float x, Float tau) {
    x = std::abs(x);
    if (x < 1e-5f) return 1;
    if (x > 1.f) return 0;
    x = (Float)(x * kPiD);
    Float s = std::sin(x * tau) / (x * tau);
    Float l = std::sin(x) / x;
    return s * l;
}

/* geometry.h:1083 Mod: remainder made non-negative */
static inline int wrap(int a, int n) {
    int r = a - (a / n) * n;
    return r < 0 ? r + n : r;
}

struct Tap4 {
    int first;
    Float w[4];
};
/* GetResampleWeights, texture.cpp:91-109 */
static std::vector<Tap4> resample_taps(int from, int to) {
    std::vector<Tap4> taps((size_t)to);
    const Float filterwidth = 2.f;
    for (int i = 0; i < to; i++) {
        Float center = (i + .5f) * from / to;
        taps[i].first = (int)std::floor((center - filterwidth) + 0.5f);
        for (int j = 0; j < 4; j++) {
            Float pos = taps[i].first + j + .5f;
            taps[i].w[j] = lanczos((pos - center) / filterwidth, 2);
        }
        Float inv = 1 / (taps[i].w[0] + taps[i].w[1] + taps[i].w[2] + taps[i].w[3]);
        for (int j = 0; j < 4; j++) taps[i].w[j] *= inv;
    }
    return taps;
}

struct Plane {
    int rx = 0, ry = 0;
    std::vector<Rgb> px;
    /* Texel with ImageWrap::Repeat, texture.cpp:68-88 / texture.h:132-153 */
    const Rgb &at(int s, int t) const { return px[(size_t)wrap(s, rx) + (size_t)wrap(t, ry) * rx]; }
};

static bool is_pow2(int v) { return v && !(v & (v - 1)); }
static int round_up_pow2(int v) { /* texture.h:102-108 */
    v--;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}
static int log2_int(uint64_t n) { /* geometry.h:99-103 */
    int i = -(n == 0);
    for (int k = 32; k >= 1; k >>= 1)
        if (n >= (UINT64_C(1) << k)) { i += k; n >>= k; }
    return i;
}

/* level 0 of BuildMipMap, texture.cpp:114-178 (wrapMode = Repeat: the in-range test of :135,:158 always holds) */
static Plane level_zero(const std::vector<Rgb> &img, int w, int h) {
    Plane out;
    if (is_pow2(w) && is_pow2(h)) {
        out.rx = w; out.ry = h; out.px = img;
        return out;
    }
    const int pw = round_up_pow2(w), ph = round_up_pow2(h);
    out.rx = pw; out.ry = ph;
    out.px.assign((size_t)pw * ph, Rgb{0.f, 0.f, 0.f});
    const std::vector<Tap4> sT = resample_taps(w, pw);
    for (int t = 0; t < h; t++)
        for (int s = 0; s < pw; s++) {
            Rgb v{0.f, 0.f, 0.f};
            for (int j = 0; j < 4; j++) v = v + sT[s].w[j] * img[(size_t)t * w + wrap(sT[s].first + j, w)];
            out.px[(size_t)t * pw + s] = v;
        }
    const std::vector<Tap4> tT = resample_taps(h, ph);
    std::vector<Rgb> column((size_t)ph);
    for (int s = 0; s < pw; s++) {
        for (int t = 0; t < ph; t++) {
            Rgb v{0.f, 0.f, 0.f};
            for (int j = 0; j < 4; j++) v = v + tT[t].w[j] * out.px[(size_t)wrap(tT[t].first + j, h) * pw + s];
            column[t] = v;
        }
        for (int t = 0; t < ph; t++) { /* Clamp(workData[t], T(0), T(1)): per component (geometry.h:568-572) */
            Rgb v = column[t];
            v.r = v.r < 0.f ? 0.f : (v.r > 1.f ? 1.f : v.r);
            v.g = v.g < 0.f ? 0.f : (v.g > 1.f ? 1.f : v.g);
            v.b = v.b < 0.f ? 0.f : (v.b > 1.f ? 1.f : v.b);
            out.px[(size_t)t * pw + s] = v;
        }
    }
    return out;
}

struct Pyramid {
    std::vector<Plane> lv;
    /* MipMap::triangle, texture.h:155-166 */
    Rgb triangle(int level, Float u, Float v) const {
        const int last = (int)lv.size() - 1;
        level = level < 0 ? 0 : (level > last ? last : level);
        const Plane &p = lv[level];
        Float s = u * p.rx - 0.5f, t = v * p.ry - 0.5f;
        int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
        Float ds = s - s0, dt = t - t0;
        return (1 - ds) * (1 - dt) * p.at(s0, t0) + (1 - ds) * dt * p.at(s0, t0 + 1) + ds * (1 - dt) * p.at(s0 + 1, t0) + ds * dt * p.at(s0 + 1, t0 + 1);
    }
    /* MipMap::Lookup(st, width), texture.h:168-179 */
    Rgb lookup(Float u, Float v, Float width) const {
        const int n = (int)lv.size();
        const Float invLog2 = 1.442695040888963387004650940071;
        Float level = n - 1 + std::log(width > (Float)1e-8 ? width : (Float)1e-8) * invLog2;
        if (level < 0) return triangle(0, u, v);
        if (level >= n - 1) return lv[n - 1].at(0, 0);
        int il = (int)std::floor(level);
        Float delta = level - il;
        return (1 - delta) * triangle(il, u, v) + delta * triangle(il + 1, u, v); /* Lerp, geometry.h:1025-1027 */
    }
};

/* CreateDistribution1D, infinite.cpp:7-22 */
static Float distribution_1d(const Float *f, int n, Float *cdf) {
    cdf[0] = 0;
    for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + f[i - 1] / n;
    const Float integral = cdf[n];
    if (integral == 0)
        for (int i = 1; i < n + 1; ++i) cdf[i] = Float(i) / Float(n);
    else
        for (int i = 1; i < n + 1; ++i) cdf[i] /= integral;
    return integral;
}

struct EnvTables {
    int width = 0, height = 0, nu = 0, nv = 0;
    std::vector<Float> texels, func, cdf, func_int, marg_cdf;
    Float marg_int = 0;
};

} /* namespace */

extern "C" {

/* rgb: height rows of width texels, 3 floats each (what ReadImageEXR hands BuildSpectrumMipMap) */
void *orc_envmap_build(const float *rgb, int width, int height, const float *scale3) {
    if (!rgb || !scale3 || width <= 0 || height <= 0) return nullptr;
    std::vector<Rgb> L((size_t)width * height);
    for (size_t i = 0; i < L.size(); i++) /* texture.cpp:224-226 */
        L[i] = Rgb{rgb[3 * i] * scale3[0], rgb[3 * i + 1] * scale3[1], rgb[3 * i + 2] * scale3[2]};
    Pyramid pyr;
    pyr.lv.push_back(level_zero(L, width, height));
    const int n_levels = 1 + log2_int((uint64_t)(pyr.lv[0].rx > pyr.lv[0].ry ? pyr.lv[0].rx : pyr.lv[0].ry));
    for (int i = 1; i < n_levels; i++) { /* texture.cpp:189-212 */
        const Plane &p = pyr.lv[i - 1];
        Plane c;
        c.rx = p.rx / 2 > 1 ? p.rx / 2 : 1;
        c.ry = p.ry / 2 > 1 ? p.ry / 2 : 1;
        c.px.resize((size_t)c.rx * c.ry);
        for (int t = 0; t < c.ry; t++)
            for (int s = 0; s < c.rx; s++)
                c.px[(size_t)s + (size_t)t * c.rx] = 0.25f * (p.at(2 * s, 2 * t) + p.at(2 * s + 1, 2 * t) + p.at(2 * s, 2 * t + 1) + p.at(2 * s + 1, 2 * t + 1));
        pyr.lv.push_back(c);
    }
    EnvTables *e = new EnvTables();
    e->width = pyr.lv[0].rx;
    e->height = pyr.lv[0].ry;
    e->texels.resize(3 * (size_t)e->width * e->height);
    for (size_t i = 0; i < pyr.lv[0].px.size(); i++) {
        e->texels[3 * i] = pyr.lv[0].px[i].r; e->texels[3 * i + 1] = pyr.lv[0].px[i].g; e->texels[3 * i + 2] = pyr.lv[0].px[i].b;
    }
    /* texture.cpp:235-252 */
    const int W = 2 * e->width, H = 2 * e->height;
    e->nu = W; e->nv = H;
    e->func.resize((size_t)W * H);
    const float fwidth = 0.5f / (W < H ? W : H);
    for (int v = 0; v < H; v++) {
        Float vp = (v + .5f) / (Float)H;
        Float sinTheta = std::sin((Float)(kPiD * (v + .5f) / H));
        for (int u = 0; u < W; u++) {
            Float up = (u + .5f) / (Float)W;
            const Rgb c = pyr.lookup(up, vp, fwidth);
            Float y = 0.212671f * c.r + 0.715160f * c.g + 0.072169f * c.b; /* Spectrum::Y, geometry.h:419-422 */
            e->func[(size_t)u + (size_t)v * W] = y;
            e->func[(size_t)u + (size_t)v * W] *= sinTheta;
        }
    }
    /* CreateDistribution2D, infinite.cpp:24-39 */
    e->cdf.resize((size_t)(W + 1) * H);
    e->func_int.resize((size_t)H);
    for (int v = 0; v < H; v++) e->func_int[v] = distribution_1d(&e->func[(size_t)v * W], W, &e->cdf[(size_t)v * (W + 1)]);
    e->marg_cdf.resize((size_t)H + 1);
    e->marg_int = distribution_1d(e->func_int.data(), H, e->marg_cdf.data());
    return e;
}
void orc_envmap_view(const void *h, int32_t *dims4, const float **texels, const float **func, const float **cdf, const float **func_int,
                     const float **marg_cdf, float *marg_int) {
    const EnvTables *e = (const EnvTables *)h;
    dims4[0] = e->width; dims4[1] = e->height; dims4[2] = e->nu; dims4[3] = e->nv;
    *texels = e->texels.data(); *func = e->func.data(); *cdf = e->cdf.data(); *func_int = e->func_int.data(); *marg_cdf = e->marg_cdf.data();
    *marg_int = e->marg_int;
}
void orc_envmap_free(void *h) { delete (EnvTables *)h; }

} /* extern "C" */
