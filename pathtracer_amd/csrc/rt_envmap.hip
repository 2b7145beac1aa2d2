/*
 * rt_envmap.hip — light-map tables built on the device (device half).
 *
 * Replaces the host pipeline the reference runs once per scene in InsertEXRLightMap (scene.cpp:703-719):
 *   BuildSpectrumMipMap  texture.cpp:111-215   scale, Lanczos resample to powers of two (wrap = Repeat, clamp to [0,1]
 *                                              in that path only), box-filter pyramid
 *   the func table       texture.cpp:235-252   luminance of a trilinear MIPMap::Lookup at (u+.5)/W, (v+.5)/H of width
 *                                              0.5/min(W,H), times sin(theta) of the row, W x H = 2 width x 2 height
 *   CreateDistribution2D infinite.cpp:7-39     a running-sum CDF per row, their integrals, the marginal CDF
 * and must give the SAME floats as the host mirror (host/pt_scene.cpp InsertLightMap), so:
 *   - everything transcendental is O(W + H) work and stays on the host side of this file with glibc: the Lanczos weights
 *     of the two resample passes, sin(theta) per row, and log(width) of the lookup (one number);
 *   - every sum keeps the reference's order: the four-tap resamples, the box filter, the bilinear / trilinear
 *     blends term by term (-ffp-contract=off), and the CDFs as sequential running sums, ONE LANE PER ROW - a parallel
 *     scan would round differently.  Rows are staged through LDS in 64 x 64 tiles so that global reads and writes
 *     are whole 256-byte rows per wavefront although each lane walks its own row.
 * All kernels are streaming passes over at most 4 W H floats: HBM-bound, a few hundred microseconds for a 2048 x 1024 map.
 */
#include "rt_common.h"
#include "ptr_host.h"
#include <cmath>
#include <vector>

namespace {

struct RWeight { /* texture.cpp:91-109 */
    int first;
    float w[4];
};
__device__ __forceinline__ int mod_i(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }

__global__ void __launch_bounds__(TPB) k_env_scale(const float *in, size_t n3, float sx, float sy, float sz, float *out) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n3) return;
    const int c = (int)(i % 3);
    out[i] = in[i] * (c == 0 ? sx : (c == 1 ? sy : sz));
}
/* texture.cpp:124-137: rows of the source resampled along s */
__global__ void __launch_bounds__(TPB) k_env_resample_s(const float *L, int rx, int ry, int px, const RWeight *sw, float *res) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= (size_t)px * ry) return;
    const int s = (int)(i % px), t = (int)(i / px);
    const RWeight w = sw[s];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int j = 0; j < 4; j++) {
        const float *p = L + 3 * ((size_t)t * rx + mod_i(w.first + j, rx));
        ax = ax + w.w[j] * p[0]; ay = ay + w.w[j] * p[1]; az = az + w.w[j] * p[2];
    }
    float *o = res + 3 * i;
    o[0] = ax; o[1] = ay; o[2] = az;
}
/* texture.cpp:139-163: columns resampled along t, clamped to [0, 1] */
__global__ void __launch_bounds__(TPB) k_env_resample_t(const float *res, int ry, int px, int py, const RWeight *tw, float *out) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= (size_t)px * py) return;
    const int s = (int)(i % px), t = (int)(i / px);
    const RWeight w = tw[t];
    float a[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < 4; j++) {
        const float *p = res + 3 * ((size_t)mod_i(w.first + j, ry) * px + s);
        a[0] = a[0] + w.w[j] * p[0]; a[1] = a[1] + w.w[j] * p[1]; a[2] = a[2] + w.w[j] * p[2];
    }
    float *o = out + 3 * i;
    for (int c = 0; c < 3; c++) o[c] = a[c] < 0 ? 0 : (a[c] > 1 ? 1 : a[c]);
}
/* texture.cpp:189-212: one level of the pyramid from the one above */
__global__ void __launch_bounds__(TPB) k_env_down(const float *p, int prx, int pry, int crx, int cry, float *c) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= (size_t)crx * cry) return;
    const int s = (int)(i % crx), t = (int)(i / crx);
    const float *t00 = p + 3 * ((size_t)mod_i(2 * s, prx) + (size_t)mod_i(2 * t, pry) * prx);
    const float *t10 = p + 3 * ((size_t)mod_i(2 * s + 1, prx) + (size_t)mod_i(2 * t, pry) * prx);
    const float *t01 = p + 3 * ((size_t)mod_i(2 * s, prx) + (size_t)mod_i(2 * t + 1, pry) * prx);
    const float *t11 = p + 3 * ((size_t)mod_i(2 * s + 1, prx) + (size_t)mod_i(2 * t + 1, pry) * prx);
    for (int k = 0; k < 3; k++) c[3 * i + k] = 0.25f * (((t00[k] + t10[k]) + t01[k]) + t11[k]);
}
struct LevelRef {
    const float *t;
    int rx, ry;
};
/* MIPMap::triangle, texture.h:118-131 */
__device__ __forceinline__ void env_triangle(const LevelRef &l, float st0, float st1, float out[3]) {
    const float s = st0 * l.rx - 0.5f, t = st1 * l.ry - 0.5f;
    const int s0 = (int)floorf(s), t0 = (int)floorf(t);
    const float ds = s - s0, dt = t - t0;
    const float *a = l.t + 3 * ((size_t)mod_i(s0, l.rx) + (size_t)mod_i(t0, l.ry) * l.rx);
    const float *b = l.t + 3 * ((size_t)mod_i(s0, l.rx) + (size_t)mod_i(t0 + 1, l.ry) * l.rx);
    const float *c = l.t + 3 * ((size_t)mod_i(s0 + 1, l.rx) + (size_t)mod_i(t0, l.ry) * l.rx);
    const float *d = l.t + 3 * ((size_t)mod_i(s0 + 1, l.rx) + (size_t)mod_i(t0 + 1, l.ry) * l.rx);
    const float wa = (1 - ds) * (1 - dt), wb = (1 - ds) * dt, wc = ds * (1 - dt), wd = ds * dt;
    for (int k = 0; k < 3; k++) out[k] = ((wa * a[k] + wb * b[k]) + wc * c[k]) + wd * d[k];
}
/* texture.cpp:235-252 with MIPMap::Lookup (texture.h:150-166): mode 0 = triangle(level 0), 1 = the last level's one
 * texel, 2 = blend of levels lo / hi by delta */
__global__ void __launch_bounds__(TPB) k_env_func(LevelRef lo, LevelRef hi, int mode, float delta, int W, int H, const float *sin_theta, float *func) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= (size_t)W * H) return;
    const int u = (int)(i % W), v = (int)(i / W);
    const float up = (u + .5f) / (float)W, vp = (v + .5f) / (float)H;
    float s[3];
    if (mode == 0) env_triangle(lo, up, vp, s);
    else if (mode == 1) { s[0] = lo.t[0]; s[1] = lo.t[1]; s[2] = lo.t[2]; }
    else {
        float a[3], b[3];
        env_triangle(lo, up, vp, a);
        env_triangle(hi, up, vp, b);
        for (int k = 0; k < 3; k++) s[k] = (1 - delta) * a[k] + delta * b[k];
    }
    float y = (0.212671f * s[0] + 0.715160f * s[1]) + 0.072169f * s[2];
    y *= sin_theta[v];
    func[i] = y;
}
/* Distribution1D (infinite.cpp:7-22) of `rows` rows of n values: cdf[r][0] = 0, cdf[r][i] = cdf[r][i-1] + f[r][i-1] / n
 * as a running sum, one lane per row; a wavefront takes 64 rows and moves 64-column tiles through LDS.  Then the
 * normalisation by the row's integral.  One block = one wavefront. */
__global__ void __launch_bounds__(64) k_env_row_cdf(const float *func, int rows, int n, float *cdf, float *func_int) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x, row0 = blockIdx.x * 64, my = row0 + lane;
    const float fn = (float)n;
    float run = 0.f;
    if (my < rows) cdf[(size_t)my * (n + 1)] = 0.f;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int nc = n - c0 < 64 ? n - c0 : 64;
        for (int r = 0; r < 64; r++)
            if (row0 + r < rows && lane < nc) tile[r][lane] = func[(size_t)(row0 + r) * n + c0 + lane];
        __syncthreads();
        if (my < rows)
            for (int j = 0; j < nc; j++) {
                run = run + tile[lane][j] / fn;
                tile[lane][j] = run;
            }
        __syncthreads();
        for (int r = 0; r < 64; r++)
            if (row0 + r < rows && lane < nc) cdf[(size_t)(row0 + r) * (n + 1) + 1 + c0 + lane] = tile[r][lane];
        __syncthreads();
    }
    if (my < rows) func_int[my] = run;
}
__global__ void __launch_bounds__(TPB) k_env_cdf_norm(float *cdf, int rows, int n, const float *func_int) {
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= (size_t)rows * n) return;
    const int r = (int)(i / n), k = (int)(i % n) + 1;
    const float fi = func_int[r];
    float *p = cdf + (size_t)r * (n + 1) + k;
    if (fi == 0) *p = (float)k / (float)n;
    else *p = *p / fi;
}

float lanczos_h(float x, float tau = 2) { /* texture.cpp:19-27 */
    x = std::abs(x);
    if (x < 1e-5f) return 1;
    if (x > 1.f) return 0;
    x *= 3.14159265358979323846; /* the reference's Pi is a double literal: float x double, narrowed */
    float s = std::sin(x * tau) / (x * tau);
    float lanczos = std::sin(x) / x;
    return s * lanczos;
}
std::vector<RWeight> resample_weights(int oldRes, int newRes) { /* texture.cpp:91-109 */
    std::vector<RWeight> wt(newRes);
    const float filterwidth = 2.f;
    for (int i = 0; i < newRes; i++) {
        float center = (i + .5f) * oldRes / newRes;
        wt[i].first = (int)std::floor((center - filterwidth) + 0.5f);
        for (int j = 0; j < 4; j++) {
            float pos = wt[i].first + j + .5f;
            wt[i].w[j] = lanczos_h((pos - center) / filterwidth);
        }
        float inv = 1 / (wt[i].w[0] + wt[i].w[1] + wt[i].w[2] + wt[i].w[3]);
        for (int j = 0; j < 4; ++j) wt[i].w[j] *= inv;
    }
    return wt;
}
bool is_pow2(int v) { return v && !(v & (v - 1)); }
int round_up_pow2(int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
int log2_int(unsigned v) { int i = -1; while (v) { v >>= 1; i++; } return i; }

} // namespace

#define ECHK(expr)                                                                     \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) { if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e_); rc = PTR_ERR_HIP; goto done; } \
    } while (0)

int ptr_gpu_build_envmap(const float *rgb, int width, int height, const float scale[3], HostEnvTables *out, double *ms, std::string *err) {
    int rc = PTR_OK;
    std::vector<void *> dev;
    auto dalloc = [&](size_t bytes) -> void * {
        void *p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
        dev.push_back(p);
        return p;
    };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto grid = [](size_t n) { return dim3((unsigned)((n + TPB - 1) / TPB)); };
    {
        const int rx = width, ry = height;
        const size_t n_in = (size_t)rx * ry;
        float *d_in = (float *)dalloc(12 * n_in), *d_scaled = (float *)dalloc(12 * n_in);
        if (!d_in || !d_scaled) { if (err) *err = "hipMalloc failed"; rc = PTR_ERR_HIP; goto done; }
        ECHK(hipMemcpy(d_in, rgb, 12 * n_in, hipMemcpyHostToDevice));
        ECHK(hipEventCreate(&e0));
        ECHK(hipEventCreate(&e1));
        ECHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_env_scale, grid(3 * n_in), dim3(TPB), 0, 0, d_in, 3 * n_in, scale[0], scale[1], scale[2], d_scaled);
        float *level0 = d_scaled;
        int lx = rx, ly = ry;
        if (!is_pow2(rx) || !is_pow2(ry)) {
            const int px = round_up_pow2(rx), py = round_up_pow2(ry);
            const std::vector<RWeight> sw = resample_weights(rx, px), tw = resample_weights(ry, py);
            RWeight *d_sw = (RWeight *)dalloc(sizeof(RWeight) * px), *d_tw = (RWeight *)dalloc(sizeof(RWeight) * py);
            float *d_res = (float *)dalloc(12 * (size_t)px * ry), *d_l0 = (float *)dalloc(12 * (size_t)px * py);
            if (!d_sw || !d_tw || !d_res || !d_l0) { if (err) *err = "hipMalloc failed"; rc = PTR_ERR_HIP; goto done; }
            ECHK(hipMemcpy(d_sw, sw.data(), sizeof(RWeight) * px, hipMemcpyHostToDevice));
            ECHK(hipMemcpy(d_tw, tw.data(), sizeof(RWeight) * py, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_env_resample_s, grid((size_t)px * ry), dim3(TPB), 0, 0, d_scaled, rx, ry, px, d_sw, d_res);
            hipLaunchKernelGGL(k_env_resample_t, grid((size_t)px * py), dim3(TPB), 0, 0, d_res, ry, px, py, d_tw, d_l0);
            level0 = d_l0; lx = px; ly = py;
        }
        /* pyramid (only the levels the lookup of the func table touches are needed, but they are a third of level 0) */
        const int n_levels = 1 + log2_int((unsigned)(lx > ly ? lx : ly));
        std::vector<LevelRef> pyr(n_levels);
        pyr[0] = LevelRef{level0, lx, ly};
        for (int i = 1; i < n_levels; i++) {
            const int cx = pyr[i - 1].rx / 2 > 1 ? pyr[i - 1].rx / 2 : 1, cy = pyr[i - 1].ry / 2 > 1 ? pyr[i - 1].ry / 2 : 1;
            float *d_c = (float *)dalloc(12 * (size_t)cx * cy);
            if (!d_c) { if (err) *err = "hipMalloc failed"; rc = PTR_ERR_HIP; goto done; }
            hipLaunchKernelGGL(k_env_down, grid((size_t)cx * cy), dim3(TPB), 0, 0, pyr[i - 1].t, pyr[i - 1].rx, pyr[i - 1].ry, cx, cy, d_c);
            pyr[i] = LevelRef{d_c, cx, cy};
        }
        const int W = 2 * lx, H = 2 * ly;
        /* MIPMap::Lookup(st, width), texture.h:150-166 */
        const float fwidth = 0.5f / (float)(W < H ? W : H);
        const float inv_log2 = 1.442695040888963387004650940071;
        const float level = n_levels - 1 + std::log(fwidth > 1e-8f ? fwidth : (float)1e-8) * inv_log2;
        int mode, ilo = 0, ihi = 0;
        float delta = 0;
        if (level < 0) mode = 0;
        else if (level >= n_levels - 1) { mode = 1; ilo = n_levels - 1; }
        else {
            mode = 2;
            ilo = (int)std::floor(level);
            ihi = ilo + 1;
            delta = level - ilo;
        }
        std::vector<float> sin_theta(H);
        for (int v = 0; v < H; v++) sin_theta[v] = std::sin((float)(3.14159265358979323846 * (v + .5f) / H));
        float *d_sin = (float *)dalloc(4 * (size_t)H), *d_func = (float *)dalloc(4 * (size_t)W * H), *d_cdf = (float *)dalloc(4 * (size_t)(W + 1) * H);
        float *d_fint = (float *)dalloc(4 * (size_t)H), *d_marg = (float *)dalloc(4 * (size_t)(H + 1)), *d_mint = (float *)dalloc(16);
        if (!d_sin || !d_func || !d_cdf || !d_fint || !d_marg || !d_mint) { if (err) *err = "hipMalloc failed"; rc = PTR_ERR_HIP; goto done; }
        ECHK(hipMemcpy(d_sin, sin_theta.data(), 4 * (size_t)H, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_env_func, grid((size_t)W * H), dim3(TPB), 0, 0, pyr[ilo], pyr[ihi], mode, delta, W, H, d_sin, d_func);
        hipLaunchKernelGGL(k_env_row_cdf, dim3((unsigned)((H + 63) / 64)), dim3(64), 0, 0, d_func, H, W, d_cdf, d_fint);
        hipLaunchKernelGGL(k_env_cdf_norm, grid((size_t)W * H), dim3(TPB), 0, 0, d_cdf, H, W, d_fint);
        /* the marginal: Distribution1D over the H row integrals */
        hipLaunchKernelGGL(k_env_row_cdf, dim3(1), dim3(64), 0, 0, d_fint, 1, H, d_marg, d_mint);
        hipLaunchKernelGGL(k_env_cdf_norm, grid((size_t)H), dim3(TPB), 0, 0, d_marg, 1, H, d_mint);
        ECHK(hipGetLastError());
        ECHK(hipEventRecord(e1, 0));
        ECHK(hipEventSynchronize(e1));
        float fms = 0;
        ECHK(hipEventElapsedTime(&fms, e0, e1));
        if (ms) *ms = fms;
        out->width = lx; out->height = ly; out->nu = W; out->nv = H;
        out->texels.resize(3 * (size_t)lx * ly);
        out->func.resize((size_t)W * H);
        out->cdf.resize((size_t)(W + 1) * H);
        out->func_int.resize(H);
        out->marg_cdf.resize(H + 1);
        ECHK(hipMemcpy(out->texels.data(), level0, 12 * (size_t)lx * ly, hipMemcpyDeviceToHost));
        ECHK(hipMemcpy(out->func.data(), d_func, 4 * (size_t)W * H, hipMemcpyDeviceToHost));
        ECHK(hipMemcpy(out->cdf.data(), d_cdf, 4 * (size_t)(W + 1) * H, hipMemcpyDeviceToHost));
        ECHK(hipMemcpy(out->func_int.data(), d_fint, 4 * (size_t)H, hipMemcpyDeviceToHost));
        ECHK(hipMemcpy(out->marg_cdf.data(), d_marg, 4 * (size_t)(H + 1), hipMemcpyDeviceToHost));
        ECHK(hipMemcpy(&out->marg_int, d_mint, 4, hipMemcpyDeviceToHost));
    }
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (void *p : dev) (void)hipFree(p);
    return rc;
}
