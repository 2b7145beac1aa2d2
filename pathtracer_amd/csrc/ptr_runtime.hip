/*
 * ptr_runtime.hip — kernels of the wavefront pipeline and the C-ABI of include/ptr_api.h.
 *
 * Reference launch sites replaced: SetupPixels<<<>>> (main.cpp:38-53), Render<<<>>>
 * (main.cpp:421-457) and the scene bring-up kernels (scene.cpp:960-987).  The reference's Render
 * is one megakernel, a thread per pixel looping over samples and bounces; here one "pass" carries
 * a batch of pixel-samples through
 *     raygen -> { trace_closest -> shade -> nee } per bounce -> film
 * with compacted queues of live paths between the stages, so the traversal kernel runs with full
 * wavefronts and small per-lane state.
 */
#include "rt_common.h"
#include "ptr_host.h"
#include <cfloat>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

PD void camera_ray(const DCamera &c, const PassParams &pp, int pixel, int sample, f3 *o, f3 *d, float *weight) {
    int i = pixel % pp.width, j = pixel / pp.width;
    Rng st = make_rng(pp.seed, (uint32_t)pixel, (uint32_t)sample);
    /* main.cpp:436-443 */
    float u0 = (float)i / (float)pp.width;
    float u1 = (float)j / (float)pp.height;
    float u = ((float)i + st.next()) / (float)pp.width;
    float v = ((float)j + st.next()) / (float)pp.height;
    float py = st.next(), px = st.next();
    f2 disk = concentric_sample_disk(f2(px, py));
    /* Camera::SpawnRay, camera.cpp:219-224 */
    f2 rdk = c.lens_radius * disk;
    f3 cu = ld3(c.u), cv = ld3(c.v), pos = ld3(c.position);
    f3 offset = cu * rdk.x + cv * rdk.y;
    f3 dir = ld3(c.lower_left) + u * ld3(c.horizontal) + v * ld3(c.vertical) - pos - offset;
    *o = pos + offset;
    *d = normalize(dir);
    /* GetFilterWeight, main.cpp:412-419 */
    float x = absr(u - u0);
    float y = absr(v - u1);
    float alpha = 2.f;
    float e = expf(-alpha);
    *weight = fmaxr(0.f, float(expf(-alpha * x * x) - e)) * fmaxr(0.f, float(expf(-alpha * y * y) - e));
}

/* ------------------------------------------------------------------------------------------
 * kernels
 * ---------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(TPB) k_raygen(DScene sc, DCamera cam, Pipe pp, PassParams prm, int n) {
    int slot = blockIdx.x * TPB + threadIdx.x;
    if (slot >= n) return;
    int pixel = pass_pixel(prm, slot);
    int sample = prm.sample_first + (slot / prm.npix) * prm.sample_step;
    f3 o, d;
    float w;
    camera_ray(cam, prm, pixel, sample, &o, &d, &w);
    int cap = pp.cap;
    pp.ro[slot] = o.x; pp.ro[cap + slot] = o.y; pp.ro[2 * cap + slot] = o.z;
    pp.rd[slot] = d.x; pp.rd[cap + slot] = d.y; pp.rd[2 * cap + slot] = d.z;
    PathRec ps;
    ps.L[0] = ps.L[1] = ps.L[2] = 0;
    ps.beta[0] = ps.beta[1] = ps.beta[2] = 1;
    ps.bounces = 0;
    ps.flags = 0;
    ps.dim = 0; /* the integrator restarts the pixel's stream: Render() hands Li a copy-by-value state, main.cpp:433-447 */
    ps.medium = sc.view_medium;
    ps.weight = w;
    ps.pad = 0;
    soa_store<0, PATH_WORDS>(pp.paths, pp.cap, slot, &ps);
}

/* GetPixelRGB (camera.cpp:36-51) with the three tone maps (camera.cpp:6-33) and GammaCorrect
 * (geometry.h:73-76): 28 B in, 3 B out per pixel */
PD float gamma_correct(float v) {
    if (v <= 0.0031308f) return 12.92f * v;
    return 1.055f * powf(v, (float)(1.f / 2.4f)) - 0.055f;
}
__global__ void __launch_bounds__(TPB) k_tonemap(const float *we, const double *acc, int npix, float exposure, int algo, unsigned char *out) {
    int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= npix) return;
    f3 e(we[3 * p], we[3 * p + 1], we[3 * p + 2]);
    const double a = acc[p];
    float invNs = (float)(1.0f / a);
    e *= invNs;
    if (algo == PTR_TONEMAP_REINHARD) {
        e = f3(e.x * (1.0f / (e.x + 1.f)), e.y * (1.0f / (e.y + 1.f)), e.z * (1.0f / (e.z + 1.f))); /* vec / vec: component-wise product with the reciprocals */
    } else if (algo == PTR_TONEMAP_EXPONENTIAL) {
        f3 v = -e * exposure;
        e = f3(1.f) - f3(expf(v.x), expf(v.y), expf(v.z));
    } else {
        const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f, W = 11.2f;
        e *= exposure;
        f3 num = e * (A * e + f3(C * B)) + f3(D * E), den = e * (A * e + f3(B)) + f3(D * F);
        e = divv(num, den) - f3(E / F);
        float white = ((W * (A * W + C * B) + D * E) / (W * (A * W + B) + D * F)) - E / F;
        e = e / white;
    }
    e = f3(gamma_correct(e.x), gamma_correct(e.y), gamma_correct(e.z));
    const double r = 255.999 * e.x, g = 255.999 * e.y, b = 255.999 * e.z;
    const bool ok = a != 0;
    out[3 * p] = ok && r == r ? (unsigned char)(int)r : 0;
    out[3 * p + 1] = ok && g == g ? (unsigned char)(int)g : 0;
    out[3 * p + 2] = ok && b == b ? (unsigned char)(int)b : 0;
}

/* ray generation alone, SoA planes of n = npix (ptr_generate_rays) */
__global__ void __launch_bounds__(TPB) k_raygen_only(DCamera cam, PassParams prm, int sample, float *o_out, float *d_out, float *w_out) {
    int p = blockIdx.x * TPB + threadIdx.x;
    int n = prm.npix;
    if (p >= n) return;
    f3 o, d;
    float w;
    camera_ray(cam, prm, p, sample, &o, &d, &w);
    o_out[p] = o.x; o_out[n + p] = o.y; o_out[2 * n + p] = o.z;
    d_out[p] = d.x; d_out[n + p] = d.y; d_out[2 * n + p] = d.z;
    if (w_out) w_out[p] = w;
}

/* Paths whose ray left the scene outside any medium end here: sky radiance if the integrator adds
 * it at this vertex (main.cpp:83-86, 124-127 / 179-181, 402-404), counters.  All other live paths are
 * compacted, in order, into the queue the shade kernel reads, so that its wavefronts hold only
 * vertices that run the surface / medium code: from the second bounce on about half of all rays
 * escape to the sky, and a shade wavefront costs the same whether 30 or 64 of its lanes hit. */
#define MISS_CHUNK 16 /* groups of 64 queue entries a wavefront handles per reservation in the output queue */
/* material types whose vertex is long to shade (two or more lobes, a microfacet transmission, a BSSRDF) */
#define HEAVY_MATERIALS ((1u << MT_GLASS) | (1u << MT_TRANSLUCENT) | (1u << MT_PLASTIC) | (1u << MT_UBER) | (1u << MT_SUBSURFACE) | (1u << MT_KDSUBSURFACE) | (1u << MT_DISNEY))
/* cls != null: the surviving vertices are compacted into TWO lists by what the shade kernel will run for them - q_hit gets the
 * light ones (matte, mirror, metal, emitters, medium vertices), q_hit_b the heavy ones - counted in cls[0], cls[1]; k_hitq_join
 * then puts the second list behind the first and sets hit_count, so that a shade wavefront holds vertices of one kind
 * (north_star: rays compacted by material and alive state with wavefront ballots). */
__global__ void __launch_bounds__(TPB) k_miss(DScene sc, Pipe pp, PassParams prm, const int *n_ptr, int n_host, const int *queue, int *q_hit,
                                              int *hit_count, unsigned long long *ctrs, int *q_hit_b, int *cls) {
    unsigned total = 0, miss = 0;
    const int n = n_ptr ? *n_ptr : n_host;
    const int lane = __lane_id();
    const int wave = threadIdx.x >> 6;
    const int cap = pp.cap;
    __shared__ int s_kept[TPB / 64][2], s_out[2];
    /* a wavefront takes MISS_CHUNK x 64 consecutive entries, and the block's four wavefronts reserve their share of the output
     * queue with ONE atomic: this kernel is fast enough for a per-group atomic on one address (half a million of them per
     * bounce, then 65 K per wavefront round, now 16 K per block round) to be what it waits for */
    for (long long bbase = (long long)blockIdx.x * TPB * MISS_CHUNK; bbase < n; bbase += (long long)gridDim.x * TPB * MISS_CHUNK) {
        const long long base = bbase + (long long)wave * 64 * MISS_CHUNK;
        int slots[MISS_CHUNK];
        unsigned long long masks[MISS_CHUNK], heavy_masks[MISS_CHUNK];
        int kept = 0, kept_heavy = 0;
#pragma unroll
        for (int c = 0; c < MISS_CHUNK; c++) {
            const long long i = base + c * 64 + lane;
            bool keep = false, heavy = false;
            int slot = 0;
            if (i < n) {
                slot = queue ? queue[i] : (int)i;
                const int hp = pp.hit[slot].x;
                const int med = (int)pp.paths[plane_at(PATH_WORDS, slot, WORD_OF(PathRec, medium))];
                keep = hp >= 0 || med >= 0;
                if (cls && hp >= 0) {
                    const int m = sc.prims[hp].material;
                    heavy = m >= 0 && ((HEAVY_MATERIALS >> sc.mats[m].type) & 1u);
                }
                if (!keep) {
                    PathRec ps;
                    soa_load<0, WORD_OF(PathRec, dim)>(pp.paths, cap, slot, &ps); /* L, beta, bounces, flags */
                    const f3 rd(pp.rd[slot], pp.rd[cap + slot], pp.rd[2 * cap + slot]);
                    f3 L = ld3(ps.L);
                    const f3 beta = ld3(ps.beta);
                    total++;
                    if (prm.integrator == PTR_LI_DIRECT) {
                        for (int k = 0; k < sc.n_lights; k++) L += light_le(sc, sc.lights[k], rd);
                    } else if (prm.integrator == PTR_LI_PATH) {
                        L = sum_lights_le(sc, rd, beta, L);
                        miss++;
                        if (ps.bounces == 10 - 1) L = f3(0.f); /* sic, main.cpp:408 */
                    } else {
                        miss++;
                        const bool beta_gone = prm.integrator == PTR_LI_VOLPATH && is_black(beta); /* main.cpp:164 */
                        if (!beta_gone && (ps.bounces == 0 || (ps.flags & PFL_SPECULAR))) L = sum_lights_le(sc, rd, beta, L);
                    }
                    st3(ps.L, L);
                    soa_store<0, 3>(pp.paths, cap, slot, &ps);
                }
            }
            slots[c] = slot;
            heavy_masks[c] = __ballot(keep && heavy);
            masks[c] = __ballot(keep && !heavy);
            kept += __popcll(masks[c]);
            kept_heavy += __popcll(heavy_masks[c]);
        }
        if (lane == 0) { s_kept[wave][0] = kept; s_kept[wave][1] = kept_heavy; }
        __syncthreads();
        if (threadIdx.x < 2) {
            int sum = 0;
            for (int w = 0; w < TPB / 64; w++) sum += s_kept[w][threadIdx.x];
            int at = 0;
            if (sum) at = threadIdx.x == 0 ? atomicAdd(cls ? cls : hit_count, sum) : atomicAdd(cls + 1, sum); /* (kept_heavy is 0 without cls) */
            s_out[threadIdx.x] = at;
        }
        __syncthreads();
        int out = s_out[0], out_b = s_out[1];
        for (int w = 0; w < wave; w++) { out += s_kept[w][0]; out_b += s_kept[w][1]; }
#pragma unroll
        for (int c = 0; c < MISS_CHUNK; c++) {
            if ((masks[c] >> lane) & 1ull) q_hit[out + __popcll(masks[c] & ((1ull << lane) - 1ull))] = slots[c];
            out += __popcll(masks[c]);
            if ((heavy_masks[c] >> lane) & 1ull) q_hit_b[out_b + __popcll(heavy_masks[c] & ((1ull << lane) - 1ull))] = slots[c];
            out_b += __popcll(heavy_masks[c]);
        }
        __syncthreads(); /* s_kept / s_out are reused by the next round */
    }
    ctr_flush(ctrs, C_TOTAL, total);
    ctr_flush(ctrs, C_MISS, miss);
}

/* second half of k_miss's two-list compaction: the heavy vertices behind the light ones, and the total for the consumers */
__global__ void __launch_bounds__(TPB) k_hitq_join(const int *cls, const int *q_b, int *q_hit, int *hit_count, int cap) {
    const int na = cls[0], nb = cls[1];
    for (long long j = (long long)blockIdx.x * TPB + threadIdx.x; j < nb; j += (long long)gridDim.x * TPB)
        if (na + j < cap) q_hit[na + j] = q_b[j]; /* (na + nb <= cap: every entry is a distinct path slot) */
    if (blockIdx.x == 0 && threadIdx.x == 0) *hit_count = na + nb;
}

/* The four queues a bounce leaves behind - surviving paths, direct-light records, shadow rays, MIS rays - from
 * the route bits k_shade wrote (rt_common.h ROUTE_*).  Like k_miss, a wavefront takes ROUTE_CHUNK x 64 entries and
 * reserves its share of each queue with one atomic per queue (lane q serves queue q).
 * The queue of surviving paths holds path slots; the other three hold direct-light RECORD numbers.  by_place: a vertex's record
 * is numbered by its place in `queue` (round 3: k_shade writes the records of consecutive queue entries side by side, whatever
 * their path slots - at later bounces the live slots are a few per 128-byte line); else by the entry itself (exit-point records). */
#define ROUTE_CHUNK 8
__global__ void __launch_bounds__(TPB) k_route(const int *n_ptr, const int *queue, const int *route, int *q_next, int *q_nee, int *q_shadow,
                                               int *q_mis, int *counts, int by_place) {
    /* one reservation per queue and BLOCK round (2048 entries): the four counters share a cache line, and at one atomic per
     * wavefront round their serialisation, not the streaming, set this kernel's time (1.5 ms for the 60 M entries of bounce 0) */
    __shared__ int s_mine[TPB / 64][4], s_at[4];
    const int n = *n_ptr;
    const int lane = __lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int *const qs[4] = {q_next, q_nee, q_shadow, q_mis};
    for (long long bbase = (long long)blockIdx.x * TPB * ROUTE_CHUNK; bbase < n; bbase += (long long)gridDim.x * TPB * ROUTE_CHUNK) {
        const long long base = bbase + (long long)wave * 64 * ROUTE_CHUNK;
        int slots[ROUTE_CHUNK], recs[ROUTE_CHUNK], bits[ROUTE_CHUNK];
#pragma unroll
        for (int c = 0; c < ROUTE_CHUNK; c++) {
            const long long i = base + c * 64 + lane;
            slots[c] = 0;
            recs[c] = 0;
            bits[c] = 0;
            if (i < n) {
                slots[c] = queue ? queue[i] : (int)i;
                recs[c] = by_place ? (int)i : slots[c];
                bits[c] = route[i]; /* (the route words are numbered like the queue's entries) */
            }
        }
        int mine = 0; /* lane q: entries this wavefront adds to queue q */
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int kept = 0;
#pragma unroll
            for (int c = 0; c < ROUTE_CHUNK; c++) kept += __popcll(__ballot((bits[c] >> q) & 1));
            if (lane == q) mine = kept;
        }
        if (lane < 4) s_mine[wave][lane] = mine;
        __syncthreads();
        if (threadIdx.x < 4) {
            int total = 0;
            for (int w = 0; w < TPB / 64; w++) total += s_mine[w][threadIdx.x];
            s_at[threadIdx.x] = total ? atomicAdd(counts + threadIdx.x, total) : 0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int out = s_at[q];
            for (int w = 0; w < wave; w++) out += s_mine[w][q];
#pragma unroll
            for (int c = 0; c < ROUTE_CHUNK; c++) {
                const unsigned long long m = __ballot((bits[c] >> q) & 1);
                if ((bits[c] >> q) & 1) qs[q][out + __popcll(m & lt)] = q == 0 ? slots[c] : recs[c];
                out += __popcll(m);
            }
        }
        __syncthreads(); /* s_mine / s_at are reused by the next round */
    }
}

/* sum of a direct-light record whose two rays went through the persistent tracer; q_hit: the bounce's queue of shaded
 * vertices, whose entry r is the path slot record r belongs to */
__global__ void __launch_bounds__(TPB) k_nee_resolve(DScene sc, Pipe pp, const int *n_ptr, int n_host, const int *queue, const int *q_hit,
                                                     unsigned long long *ctrs, int shadow_closest) {
    NeeCtr ctr;
    ctr.shadow = ctr.mis = ctr.zero = 0;
    GRID_STRIDE_BEGIN(n_ptr, n_host)
    if (i < n) {
        const int r = queue[i];
        const int slot = q_hit[r];
        NeeRec rec;
        nee_load(pp, r, &rec);
        soa_load<NEE_RES_BEGIN, NEE_RES_END>(pp.nee, pp.cap, r, &rec);
        f3 Ld = nee_resolve_traced(sc, rec, ctr, shadow_closest != 0);
        PathRec ps;
        soa_load<0, 3>(pp.paths, pp.cap, slot, &ps);
        f3 L = ld3(ps.L);
        L += Ld;
        if (rec.flags & NEE_EXTRA) L += ld3(rec.extra);
        st3(ps.L, L);
        soa_store<0, 3>(pp.paths, pp.cap, slot, &ps);
    }
    GRID_STRIDE_END
    ctr_flush(ctrs, C_SHADOW, ctr.shadow);
    ctr_flush(ctrs, C_MIS, ctr.mis);
    ctr_flush(ctrs, C_ZERO, ctr.zero);
}

/* The same for the direct light at BSSRDF exit points (dev_bssrdf.h ExitNee): records numbered like the probe chains, each added
 * to the L of its chain's path slot.  Runs after k_nee_resolve, so the term follows the entry vertex's own (main.cpp:207-228). */
__global__ void __launch_bounds__(TPB) k_nee_resolve2(DScene sc, Pipe pp, SssChains ch, const int *n_ptr, const int *queue, unsigned long long *ctrs,
                                                      int shadow_closest) {
    NeeCtr ctr;
    ctr.shadow = ctr.mis = ctr.zero = 0;
    Pipe p2 = pp;
    p2.nee = reinterpret_cast<uint32_t *>(ch.exit.planes);
    GRID_STRIDE_BEGIN(n_ptr, 0)
    if (i < n) {
        const int j = queue[i];
        NeeRec rec;
        nee_load(p2, j, &rec);
        soa_load<NEE_RES_BEGIN, NEE_RES_END>(p2.nee, p2.cap, j, &rec);
        const f3 Ld = nee_resolve_traced(sc, rec, ctr, shadow_closest != 0);
        const int slot = ChainRef{ch.planes, j}.geti(SW_SLOT);
        PathRec ps;
        soa_load<0, 3>(pp.paths, pp.cap, slot, &ps);
        f3 L = ld3(ps.L);
        L += Ld;
        st3(ps.L, L);
        soa_store<0, 3>(pp.paths, pp.cap, slot, &ps);
    }
    GRID_STRIDE_END
    ctr_flush(ctrs, C_SHADOW, ctr.shadow);
    ctr_flush(ctrs, C_MIS, ctr.mis);
    ctr_flush(ctrs, C_ZERO, ctr.zero);
}

/* Render()'s accumulation (main.cpp:449-455): we += weight * L per sample, in sample order */
__global__ void __launch_bounds__(TPB) k_film(Pipe pp, int npix, const int *pix_map, int nbatch, float *we, double *acc, long long *samples) {
    int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= npix) return;
    const int lp = p; /* pixel of the pass */
    p = pix_map ? pix_map[lp] : lp; /* pixel of the film */
    f3 out(we[3 * p], we[3 * p + 1], we[3 * p + 2]);
    double a = acc[p];
    for (int b = 0; b < nbatch; b++) {
        PathRec ps;
        const int slot = b * npix + lp;
        soa_load<0, 3>(pp.paths, pp.cap, slot, &ps);
        soa_load<WORD_OF(PathRec, weight), WORD_OF(PathRec, weight) + 1>(pp.paths, pp.cap, slot, &ps);
        out += ps.weight * ld3(ps.L);
        a += ps.weight;
    }
    we[3 * p] = out.x; we[3 * p + 1] = out.y; we[3 * p + 2] = out.z;
    acc[p] = a;
    samples[p] += nbatch;
}

/* ---- gathering a tile-split film (ptr_film_tiles_pack / _unpack, include/ptr_api.h; no counterpart in the reference) ----
 * A shard (offset, stride) owns the 8x8 tiles t = offset + k * stride of the film.  Its packed film holds those tiles back
 * to back, 64 pixels each whatever part of a tile lies inside the frame (the rest is zero), so that a pixel's place follows
 * from its coordinates alone: entry k * 64 + (y & 7) * 8 + (x & 7).  Streaming copies, 28 B per pixel each way. */
__global__ void __launch_bounds__(TPB) k_film_pack_tiles(const float *we, const double *acc, const long long *smp, int width, int height, int offset,
                                                         int stride, long long n_entries, float *p_we, double *p_acc, long long *p_smp) {
    const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
    if (e >= n_entries) return;
    const int tx = (width + PTR_TILE - 1) / PTR_TILE, ty = (height + PTR_TILE - 1) / PTR_TILE;
    const long long t = offset + (e >> 6) * stride;
    const int x = (int)(t % tx) * PTR_TILE + (int)(e & 7), y = (int)(t / tx) * PTR_TILE + (int)((e >> 3) & 7);
    const bool in = t < (long long)tx * ty && x < width && y < height;
    const long long p = in ? (long long)y * width + x : 0;
    p_we[3 * e] = in ? we[3 * p] : 0.f; p_we[3 * e + 1] = in ? we[3 * p + 1] : 0.f; p_we[3 * e + 2] = in ? we[3 * p + 2] : 0.f;
    p_acc[e] = in ? acc[p] : 0.0;
    p_smp[e] = in ? smp[p] : 0;
}
__global__ void __launch_bounds__(TPB) k_film_unpack_tiles(const float *p_we, const double *p_acc, const long long *p_smp, int width, int height, int offset,
                                                           int stride, long long n_entries, float *we, double *acc, long long *smp) {
    const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
    if (e >= n_entries) return;
    const int tx = (width + PTR_TILE - 1) / PTR_TILE, ty = (height + PTR_TILE - 1) / PTR_TILE;
    const long long t = offset + (e >> 6) * stride;
    const int x = (int)(t % tx) * PTR_TILE + (int)(e & 7), y = (int)(t / tx) * PTR_TILE + (int)((e >> 3) & 7);
    if (t >= (long long)tx * ty || x >= width || y >= height) return;
    const long long p = (long long)y * width + x;
    we[3 * p] = p_we[3 * e]; we[3 * p + 1] = p_we[3 * e + 1]; we[3 * p + 2] = p_we[3 * e + 2];
    acc[p] = p_acc[e];
    smp[p] = p_smp[e];
}
__global__ void __launch_bounds__(TPB) k_fill_i64(long long *dst, long long n, long long value) {
    const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
    if (i < n) dst[i] = value;
}

/* ------------------------------------------------------------------------------------------
 * scratch of the entry point that allocates nothing (ptr_trace_closest_noalloc, include/ptr_api.h)
 *
 * The persistent tracer hands out rays through a fetch cursor in device memory.  ptr_trace_closest takes its
 * cursor from a ring the context allocated with hipMalloc; the no-allocation variant keeps it here, in the code
 * object itself.  Every GPU the library runs on holds its own copy of this array until the process ends.  In host
 * code `scratch` is only a handle: the device address is asked for once per device (ptr_create, with that device
 * current) and kept in g_scratch_ptr[device]; kernels, memsets and copies only ever see that address.
 * ---------------------------------------------------------------------------------------- */
#define PTR_SCRATCH_INTS 64 /* 256 B: word 0 = ray-fetch cursor of the call in flight, the rest reserved */
static __device__ __align__(256) int scratch[PTR_SCRATCH_INTS];
#define PTR_MAX_DEVICES 64
static int *g_scratch_ptr[PTR_MAX_DEVICES]; /* per device: where that GPU's copy of scratch[] lives */
static std::mutex g_scratch_mu;              /* ptr_create may be called from several host threads at once */

/* ------------------------------------------------------------------------------------------
 * host runtime
 * ---------------------------------------------------------------------------------------- */
static thread_local std::string g_err;
static int set_err(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                                            \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return set_err(PTR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

enum { K_RAYGEN = 0, K_TRACE, K_SHADE, K_SHADOW, K_MIS, K_RESOLVE, K_FILM, K_FINISH, K_MISS, K_MIS_ANY, K_PROBE, K_FALLBACK, K_API_TRACE = K_TRACE };
static const char *kKernelNames[PTR_PROF_KERNELS] = {"raygen", "trace_closest", "shade", "nee_shadow", "nee_mis", "nee_resolve", "film",
                                                     "finish", "miss", "nee_mis_any", "sss_probe", "trace_fallback"};

struct ProfSpan {
    hipEvent_t a, b;
    int family;
    uint64_t items;
};

/* A pass (a batch of samples) owns one lane: its own stream, path-state buffers and queues.
 * Ray cost is heavy-tailed - a launch over a few hundred late-bounce rays still lasts as long as
 * its longest ray, ~1 ms of dependent loads - so several passes are kept in flight and the tail
 * of one overlaps the bulk of the others. */
#define N_LANES 6 /* passes in flight at most (PTR_LANES; each on its own stream) */
/* ... and by default.  Six in rounds 1-3 (4 -> 6 gave +3.5 % then); since the tracer's launches stopped queueing up on their fetch
 * cursor (dev_trace.h PT_POOL) three do as much - configs[1] 2321 Mrays/s with six, 2328 with three; configs[3] 2903 / 3160,
 * configs[5] 6149 / 6416, configs[0] and [4] equal - and only they get pass buffers: 59 GB at 1080p instead of 118. */
#define PT_LANES_DEFAULT 3
#define MAX_ITERS 20
/* the fetch cursors of the pipeline's trace launches: three launches per bounce, PT_N_CURSORS cursors each on cache lines of their own */
#ifndef PT_N_CURSORS
#define PT_N_CURSORS 16
#endif
static int g_n_cursors = PT_N_CURSORS; /* PTR_CURSORS=1..PT_N_CURSORS (read by ptr_create): how many of them the launches use */
#define LANE_CURSOR_BASE ((14 * MAX_ITERS + 15) / 16 * 16)
#define LANE_CURSOR_INTS (PT_N_CURSORS * PT_CURSOR_STRIDE)
#define LANE_COUNTER_INTS (LANE_CURSOR_BASE + 3 * MAX_ITERS * LANE_CURSOR_INTS)
#define LANE_FB_COUNTS (10 * MAX_ITERS) /* fallback launches (fast top level, dev_trace.h): per bounce {count, cursor} of the path segments, then of the MIS rays */
/* BSSRDF probe chains as tracer rays (rt_shade.h): the lane's counters for them, per staged bounce: [0] deferred vertices,
 * [1] chains that got a record, [2] the ray-fetch cursor of the chain launch, [4..8) k_route's counters over the exit-point
 * records (unused, records, shadow rays, MIS rays), [8], [9] the fetch cursors of their rays */
#define SSS_EXIT_COUNTS 4
#define SSS_COUNTER_STRIDE 16
#define SSS_COUNTER_INTS (SSS_COUNTER_STRIDE * MAX_ITERS)
struct Lane {
    hipStream_t stream = nullptr;
    Pipe pipe;
    int *queues[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; /* ping, pong, nee, shadow, mis, binned */
    int *counts = nullptr;      /* device: [MAX_ITERS][4] = next, nee, shadow, mis; [MAX_ITERS][3] ray-fetch cursors;
                                   [MAX_ITERS] live paths that did not end in the miss pre-pass; [MAX_ITERS][2] of them light / heavy to shade */
    int *counts_host = nullptr; /* pinned: the [MAX_ITERS][4] counts of the last finished pass */
    int *fb_queue = nullptr;    /* closest-hit rays a fast-top-level launch leaves to the reference's tree (dev_trace.h): cap slots, used by one launch at a time */
    hipEvent_t film_done = nullptr, joined = nullptr;
    int staged_used = 0, slots_used = 0; /* of the pass last queued on this lane */
    /* probe-chain records (scenes with a BSSRDF material only; allocated at the first such pass) */
    uint32_t *sss_planes = nullptr;
    int *sss_counts = nullptr;
    int sss_cap = 0;
    /* direct light at the exit points as records of their own (dev_bssrdf.h ExitNee): planes, flags per chain, the three queues */
    float *sss_nee2 = nullptr;
    int *sss_route = nullptr, *sss_q[3] = {nullptr, nullptr, nullptr};
};

static void lane_free_sss(Lane &l) {
    for (void *p : {(void *)l.sss_planes, (void *)l.sss_counts, (void *)l.sss_nee2, (void *)l.sss_route, (void *)l.sss_q[0], (void *)l.sss_q[1], (void *)l.sss_q[2]})
        if (p) (void)hipFree(p);
    l.sss_planes = nullptr; l.sss_counts = nullptr;
    l.sss_nee2 = nullptr; l.sss_route = nullptr; l.sss_q[0] = l.sss_q[1] = l.sss_q[2] = nullptr;
}

struct PtrContext {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<void *> scene_allocs;
    DScene sc;
    bool has_scene = false, has_camera = false, has_film = false;
    bool scene_has_media = false, scene_all_opaque = false, scene_passthrough = false;
    /* one tracer launch per bounce for {next segments, shadow rays, MIS rays} instead of three: 6 % less GPU
     * time for a pass that has the GPU to itself, 2 % more when several passes overlap (bench.py): off by default */
    bool merge_traces = false;
    bool bin_hits = false;       /* PTR_BIN_HITS=1: k_miss compacts the vertices to shade into light and heavy materials (measured: no gain) */
    bool nee_one_launch = false; /* shadow + any-hit MIS rays of a bounce in one launch (PTR_NEE_ONE) */
    bool nee_closest_ok = true; /* PTR_NEE_CLOSEST=0: keep the in-kernel walks wherever a primitive has no material */
    /* PTR_FAST_TOP=1: rays walk the fast top-level tree where the scene has one and their answer cannot depend on the reference's
     * order (dev_trace.h).  Exact (films and hits bit-identical) but measured SLOWER, so off: the room of configs[2] tests 72.2 -> 62.0
     * boxes per closest-hit ray in the SAH tree - a ray through a furnished room really does cross that many primitive bounds -
     * while the checks cost the closest-hit tracer 18.3 -> 26.0 ms per launch (profiles/r03_fast_top_level.txt). */
    bool fast_top = false;
    bool mis_any_hit = true; /* MIS rays as any-hit rays where the result cannot tell the difference (PTR_MIS_ANY=0: never) */
    int scene_max_lobes = MAX_LOBES;
    bool scene_has_sss = false; /* some material carries a BSSRDF: k_shade<true> / k_finish<true> */
    int sss_rounds = 1;         /* probe chains as rays of the continuing tracer (PTR_SSS_CHAINS=0: walked in the tail kernel's lanes as in round 2) */
    int sss_cap_div = 4;        /* chain records per lane = path slots / this (PTR_SSS_CAP_DIV) */
    bool sss_exit_records = true; /* direct light at the exit points through the tracer's launches (PTR_SSS_EXIT=0: resolved in the tail kernel) */
    /* Which persistent tracer a launch uses.  One ray per lane (dev_trace.h) has the higher throughput once the GPU is full
     * (r02, 14.4 M incoherent rays: 3.8 ms against 5.7 ms); four lanes per ray (dev_qtrace.h) has a quarter of the rays in
     * flight, a third of the instructions per step and no scratch, and ends a small launch sooner (1.8 M rays: 1.04 ms
     * against 1.41 ms).  PTR_TRACER=lane (default) | quad | auto: quad for launches expected to carry at most quad_below rays.  With six passes overlapping the
     * small launches of one pass run beside the big ones of another, and auto gains nothing end to end. */
    int tracer_mode = 0; /* 0 lane (default: r02 bench 1660 Mrays/s, auto 1653, quad 1062), 1 quad, 2 auto */
    long long quad_below = 3000000;
    DCamera cam;
    int64_t bvh_bytes = 0;
    PtrBvhBuildInfo bvh_build = {};
    /* film */
    int width = 0, height = 0, npix = 0;
    uint64_t seed = 0;
    float *we = nullptr;
    double *acc = nullptr;
    long long *samples = nullptr;
    int sample_cursor = 0, shard_offset = 0, shard_stride = 1;
    /* tile sharding (ptr_film_tiles): the pixels of this context's 8x8 tiles, in tile order; null = the whole film */
    int *pix_map = nullptr;
    int npix_local = 0, tile_offset = 0, tile_stride = 1;
    unsigned char *rgb8 = nullptr; /* 3 * npix bytes: where ptr_film_rgb8 resolves the film before the copy to the host */
    hipEvent_t entry = nullptr;    /* recorded on the caller's stream when ptr_render is entered: the lanes wait for it */
    bool debug_sync = false;       /* PTR_DEBUG_SYNC=1: wait and check for errors after every stage of a pass */
    /* pipeline */
    Lane lanes[N_LANES];
    int next_lane = 0, last_film_lane = -1;
    int lanes_alloc = PT_LANES_DEFAULT; /* lanes that get pass buffers (PTR_LANES=1..N_LANES) */
    int lanes_used = PT_LANES_DEFAULT; /* passes kept in flight: lanes_alloc, or fewer, chosen per scene at upload */
    bool lanes_forced = false;
    int staged_bounces = 6; /* wavefront bounces before k_finish; adapted from the live counts of finished passes */
    bool staged_fixed = false;
    int stage_div = 1024; /* a bounce is staged while more than slots / stage_div paths are alive (PTR_STAGE_DIV) */
    int spread = 1; /* a ptr_render call deals its samples to this many passes (lanes) */
    int *api_fetch = nullptr; /* ray-fetch cursors of the stand-alone trace entry points (rotating) */
    int *api_fb = nullptr;    /* their fallback queue (fast top level): two counters, then api_fb_cap slots */
    int api_fb_cap = 0;
    int *static_scratch = nullptr; /* this device's copy of scratch[] (ptr_trace_closest_noalloc) */
    int api_fetch_next = 0;
    int persist_blocks = 1024;
    bool persist_blocks_forced = false;
    int n_cus = 256;
    int trace_waves_forced = 0; /* PTR_TRACE_WAVES=4|5: the tracer build for every scene */
    bool trace_lean = true; /* PTR_TRACE_LEAN=0: every scene gets the tracer build with the SDF shapes' code */
    int wide_from = 4;          /* analytic primitives from which a scene gets the 128-register build (PTR_TRACE_WIDE_FROM) */
    int nb_max = 1;
    unsigned long long *ctrs = nullptr;
    uint64_t camera_rays = 0;
    /* profiling */
    bool profiling = false, count_traversal = false;
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[PTR_PROF_KERNELS];
    uint64_t prof_launches[PTR_PROF_KERNELS], prof_items[PTR_PROF_KERNELS];
};

static int ctx_use(PtrContext *ctx) {
    if (!ctx) return set_err(PTR_ERR_INVALID, "null context");
    HIPCHK(hipSetDevice(ctx->device));
    return PTR_OK;
}
/* everything this context has in flight, on its lanes and on the caller's stream */
static int sync_all(PtrContext *ctx) {
    for (Lane &l : ctx->lanes)
        if (l.stream) HIPCHK(hipStreamSynchronize(l.stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PTR_OK;
}

template <typename T> static int upload(PtrContext *ctx, const T *src, size_t n, const T **out) {
    *out = nullptr;
    if (n == 0) return PTR_OK;
    void *d = nullptr;
    HIPCHK(hipMalloc(&d, n * sizeof(T)));
    ctx->scene_allocs.push_back(d);
    HIPCHK(hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T *)d;
    return PTR_OK;
}
static void free_scene(PtrContext *ctx) {
    for (void *p : ctx->scene_allocs) (void)hipFree(p);
    ctx->scene_allocs.clear();
    ctx->has_scene = false;
}
static void free_film(PtrContext *ctx) {
    void *film[] = {ctx->we, ctx->acc, ctx->samples, ctx->rgb8};
    for (void *p : film)
        if (p) (void)hipFree(p);
    ctx->we = nullptr; ctx->acc = nullptr; ctx->samples = nullptr; ctx->rgb8 = nullptr;
    if (ctx->pix_map) (void)hipFree(ctx->pix_map);
    ctx->pix_map = nullptr;
    if (!ctx->staged_fixed) ctx->staged_bounces = 6; /* the feedback below belongs to the buffers that go away */
    for (Lane &l : ctx->lanes) {
        void *ptrs[] = {l.pipe.ro, l.pipe.rd, l.pipe.hit, l.pipe.route, l.pipe.paths, l.pipe.nee,
                        l.queues[0], l.queues[1], l.queues[2], l.queues[3], l.queues[4], l.queues[5], l.counts, l.fb_queue};
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
        if (l.counts_host) (void)hipHostFree(l.counts_host);
        lane_free_sss(l);
        l.sss_cap = 0;
        memset(&l.pipe, 0, sizeof(l.pipe));
        for (int q = 0; q < 6; q++) l.queues[q] = nullptr;
        l.counts = nullptr;
        l.counts_host = nullptr;
        l.fb_queue = nullptr;
        l.staged_used = 0;
        l.slots_used = 0;
    }
    ctx->has_film = false;
}

struct ProfScope {
    PtrContext *ctx;
    hipStream_t stream;
    bool on;
    ProfSpan sp;
    ProfScope(PtrContext *c, hipStream_t s, int family, uint64_t items) : ctx(c), stream(s), on(c->profiling) {
        if (!on) return;
        sp.family = family;
        sp.items = items;
        for (hipEvent_t *e : {&sp.a, &sp.b}) {
            if (!ctx->event_pool.empty()) { *e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
            else (void)hipEventCreate(e);
        }
        (void)hipEventRecord(sp.a, stream);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(sp.b, stream);
        ctx->spans.push_back(sp);
    }
};
static void prof_drain(PtrContext *ctx) {
    for (ProfSpan &sp : ctx->spans) {
        (void)hipEventSynchronize(sp.b);
        float ms = 0;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            ctx->prof_ms[sp.family] += ms;
            ctx->prof_launches[sp.family] += 1;
            ctx->prof_items[sp.family] += sp.items;
        }
        ctx->event_pool.push_back(sp.a);
        ctx->event_pool.push_back(sp.b);
    }
    ctx->spans.clear();
}

static inline unsigned nblocks(int n) { return (unsigned)((n + TPB - 1) / TPB); }
static inline unsigned clampu(unsigned v, unsigned lo, unsigned hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* blocks of a persistent tracer launch over at most n rays: a 256-thread block holds 256 rays, or 64 with four lanes per ray */
static inline unsigned trace_blocks(const struct PtrContext *ctx, long long n);
static inline bool use_quad(const struct PtrContext *ctx, long long n);

static inline bool use_quad(const PtrContext *ctx, long long n) { return ctx->tracer_mode == 1 || (ctx->tracer_mode == 2 && n <= ctx->quad_below); }
static inline unsigned trace_blocks(const PtrContext *ctx, long long n) {
    const long long per_block = use_quad(ctx, n) ? TPB / 4 : TPB;
    long long b = (n + per_block - 1) / per_block;
    return clampu((unsigned)(b > 0x7fffffff ? 0x7fffffff : b), 1, (unsigned)ctx->persist_blocks);
}
#define API_FETCH_SLOTS 256
/* one launch of the persistent tracer; n_upper bounds the ray count for sizing the grid (the
 * count itself may be in device memory, io.n_ptr); io.counter must be zero when the kernel starts */
/* fb_queue / fb_counters ({count, fetch cursor}, both zero): a closest-hit launch in a scene with a fast top-level tree walks that
 * tree and leaves the rays whose answer may need the reference's order in fb_queue; a second launch over that queue walks the
 * reference's tree (dev_trace.h).  Without them a closest-hit launch walks the reference's tree. */
static void launch_ptrace(PtrContext *ctx, hipStream_t s, const TraceIO &io_in, int n_upper, bool any, bool count, int family, long long n_est = -1,
                          int *fb_queue = nullptr, int *fb_counters = nullptr) {
    if (n_est < 0 || n_est > n_upper) n_est = n_upper; /* what picks the tracer: the expected ray count (n_upper only bounds it) */
    const bool quad = use_quad(ctx, n_est);
    unsigned blocks = trace_blocks(ctx, quad ? (n_est > 0 ? n_est : 1) : n_upper);
    TraceIO io = io_in;
    const bool two = !any && !quad && ctx->sc.top_fast != 0 && fb_queue && fb_counters;
    io.fb_queue = two ? fb_queue : nullptr;
    io.fb_count = two ? fb_counters : nullptr;
    {
        ProfScope ps(ctx, s, family, (uint64_t)(io.n_ptr ? 0 : io.n));
        if (quad) rt_launch_qtrace(any, count, blocks, s, ctx->sc, io, ctx->ctrs);
        else rt_launch_ptrace(any, count, blocks, s, ctx->sc, io, ctx->ctrs);
    }
    if (two) {
        TraceIO io2 = io_in;
        io2.queue = fb_queue;
        io2.n_ptr = fb_counters;
        io2.n = 0;
        io2.counter = fb_counters + 1;
        io2.n_cursors = 0; /* (one cursor: the fallback queue is short) */
        io2.fb_queue = nullptr;
        io2.fb_count = nullptr;
        io2.is_fallback = 1;
        ProfScope ps(ctx, s, K_FALLBACK, 0);
        rt_launch_ptrace(false, count, trace_blocks(ctx, n_upper / 8 + 1), s, ctx->sc, io2, ctx->ctrs);
    }
}
static TraceIO soa_io(int n, const int *queue, long long plane, const float *o, const float *d, const float *tmax, int *prim, int *face,
                      float *t, float *b0, float *b1) {
    TraceIO io;
    memset(&io, 0, sizeof(io));
    io.queue = queue;
    io.n = n;
    io.o = (const char *)o; io.o_es = 4; io.o_cs = 4 * plane;
    io.d = (const char *)d; io.d_es = 4; io.d_cs = 4 * plane;
    io.tmax = (const char *)tmax; io.tmax_es = 4; io.tmax_const = FLT_MAX;
    io.prim = (char *)prim; io.face = (char *)face; io.t = (char *)t; io.b0 = (char *)b0; io.b1 = (char *)b1;
    io.prim_es = io.face_es = io.t_es = io.b0_es = io.b1_es = 4;
    return io;
}
static int api_trace(PtrContext *ctx, TraceIO io, bool any) {
    io.counter = ctx->api_fetch + (ctx->api_fetch_next++ % API_FETCH_SLOTS);
    HIPCHK(hipMemsetAsync(io.counter, 0, sizeof(int), ctx->stream));
    int *fbq = nullptr, *fbc = nullptr;
    if (!any && ctx->sc.top_fast) { /* closest hit through the fast top-level tree: a fallback queue of n slots (grown when a batch needs more) */
        if (ctx->api_fb_cap < io.n) {
            HIPCHK(hipStreamSynchronize(ctx->stream)); /* an earlier batch may still be using the old one */
            if (ctx->api_fb) (void)hipFree(ctx->api_fb);
            ctx->api_fb = nullptr;
            ctx->api_fb_cap = 0;
            HIPCHK(hipMalloc((void **)&ctx->api_fb, ((size_t)io.n + 2) * sizeof(int)));
            ctx->api_fb_cap = io.n;
        }
        fbc = ctx->api_fb; /* {count, cursor}, then the queue */
        fbq = ctx->api_fb + 2;
        HIPCHK(hipMemsetAsync(fbc, 0, 2 * sizeof(int), ctx->stream));
    }
    launch_ptrace(ctx, ctx->stream, io, io.n, any, !any && ctx->count_traversal, K_API_TRACE, -1, fbq, fbc);
    HIPCHK(hipGetLastError());
    return PTR_OK;
}

/* The stages of one bounce on a lane.  Counter block `cslot` of the lane ([0] live paths after
 * the bounce, [1] direct-light records, [2] shadow rays, [3] MIS rays) and its fetch cursors must
 * be zero; n_ptr / n_host: number of live paths entering the bounce (device counter of the
 * previous bounce, or a host value); n_upper: an upper bound used only to size grids. */
static TraceIO path_io(const Lane &l, const int *queue, const int *n_ptr, int n_host, int *cursor) {
    TraceIO io = soa_io(n_host, queue, l.pipe.cap, l.pipe.ro, l.pipe.rd, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    io.hit16 = l.pipe.hit;
    io.n_ptr = n_ptr;
    io.counter = cursor;
    io.n_cursors = cursor ? g_n_cursors : 0;
    return io;
}
/* any hit within (0, 1 - ShadowEpsilon); closest = true: the first hit (primitive into occA, distance into tA) */
static TraceIO shadow_io(const Lane &l, const int *queue, const int *n_ptr, int *cursor, bool closest = false) {
    TraceIO io;
    memset(&io, 0, sizeof(io));
    io.queue = queue; io.n_ptr = n_ptr;
    io.tile_words = NEE_WORDS;
    io.o = (const char *)l.pipe.nee; io.w_o = WORD_OF(NeeRec, oA); io.w_d = WORD_OF(NeeRec, dA);
    io.tmax_const = 1 - 0.0001f;
    io.prim = (char *)l.pipe.nee; io.w_prim = WORD_OF(NeeRec, occA); io.w_t = closest ? WORD_OF(NeeRec, tA) : -1;
    io.counter = cursor;
    io.n_cursors = cursor ? g_n_cursors : 0;
    return io;
}
static TraceIO mis_io(const Lane &l, const int *queue, const int *n_ptr, int *cursor) { /* closest hit */
    TraceIO io;
    memset(&io, 0, sizeof(io));
    io.queue = queue; io.n_ptr = n_ptr;
    io.tile_words = NEE_WORDS;
    io.o = (const char *)l.pipe.nee; io.w_o = WORD_OF(NeeRec, oB); io.w_d = WORD_OF(NeeRec, dB);
    io.tmax_const = FLT_MAX;
    io.prim = (char *)l.pipe.nee; io.w_prim = WORD_OF(NeeRec, hitB); io.w_t = WORD_OF(NeeRec, tB);
    io.counter = cursor;
    io.n_cursors = cursor ? g_n_cursors : 0;
    return io;
}
/* Direct light at the exit points as records of the tracer's launches?  Not where its rays may pass through primitives (the
 * in-kernel walks: Li_VolPath in a scene that is not all opaque), nor with the merged three-kind launch (PTR_MERGE). */
static bool exit_records(const PtrContext *ctx, const Lane &l, int integrator) {
    return ctx->scene_has_sss && ctx->sss_rounds > 0 && l.sss_planes && l.sss_nee2 && !ctx->merge_traces &&
           !(integrator == PTR_LI_VOLPATH && !ctx->scene_all_opaque);
}
/* the rays of the exit-point records (queue / count written by the second k_route of the bounce) */
static TraceIO exit_io(const Lane &l, int cslot, bool mis) {
    int *cnt2 = l.sss_counts + SSS_COUNTER_STRIDE * cslot + SSS_EXIT_COUNTS;
    TraceIO io;
    memset(&io, 0, sizeof(io));
    io.queue = l.sss_q[mis ? 2 : 1]; io.n_ptr = cnt2 + (mis ? 3 : 2);
    io.tile_words = NEE_WORDS;
    io.o = (const char *)l.sss_nee2; io.w_o = mis ? WORD_OF(NeeRec, oB) : WORD_OF(NeeRec, oA); io.w_d = mis ? WORD_OF(NeeRec, dB) : WORD_OF(NeeRec, dA);
    io.tmax_const = mis ? FLT_MAX : 1 - 0.0001f;
    io.prim = (char *)l.sss_nee2; io.w_prim = mis ? WORD_OF(NeeRec, hitB) : WORD_OF(NeeRec, occA); io.w_t = mis ? WORD_OF(NeeRec, tB) : -1;
    io.counter = cnt2 + (mis ? 5 : 4);
    return io;
}
/* The probe chains of the bounce's deferred vertices as rays of the persistent tracer (rt_shade.h, dev_bssrdf.h): the begin
 * kernel numbers the chains and leaves their first segments, one launch of the continuing tracer runs every chain to its end.
 * a.chains is filled in for the tail kernels that follow.  The lane's SSS counters must be zero (run_pass). */
static void enqueue_probe_chains(PtrContext *ctx, Lane &l, ShadeArgs &a, int n_upper, int cslot, unsigned tail_blocks) {
    hipStream_t s = l.stream;
    int *cnt = l.sss_counts + SSS_COUNTER_STRIDE * cslot;
    a.chains = SssChains{l.sss_planes, l.sss_cap, cnt + 0, cnt + 1, ExitNee{exit_records(ctx, l, a.prm.integrator) ? l.sss_nee2 : nullptr, l.sss_route}};
    rt_launch_shade_begin(tail_blocks, s, a);
    const int upper = n_upper < l.sss_cap ? n_upper : l.sss_cap;
    ProfScope ps(ctx, s, K_PROBE, 0);
    rt_launch_pchain(clampu(nblocks(upper), 1, (unsigned)ctx->persist_blocks), s, ctx->sc, l.sss_planes, cnt + 1, cnt + 2, ctx->ctrs);
}
static void enqueue_shade(PtrContext *ctx, Lane &l, const PassParams &prm, const int *n_ptr, int n_host, int n_upper, const int *q_cur, int *q_next,
                          int cslot) {
    hipStream_t s = l.stream;
    int *cnt = l.counts + 4 * cslot;
    const unsigned stride_blocks = clampu(nblocks(n_upper), 32, 2048);
    int *hit_count = l.counts + 7 * MAX_ITERS + cslot;
    {
        ProfScope ps(ctx, s, K_MISS, 0);
        /* two lists by shading cost, joined (the direct-light queue is free between a bounce's resolve and the next bounce's route) */
        int *cls = ctx->bin_hits ? l.counts + 8 * MAX_ITERS + 2 * cslot : nullptr;
        hipLaunchKernelGGL(k_miss, dim3(stride_blocks), dim3(TPB), 0, s, ctx->sc, l.pipe, prm, n_ptr, n_host, q_cur, l.queues[5], hit_count, ctx->ctrs,
                           l.queues[2], cls);
        if (cls) hipLaunchKernelGGL(k_hitq_join, dim3(clampu(nblocks(n_upper), 32, 1024)), dim3(TPB), 0, s, (const int *)cls, (const int *)l.queues[2],
                                    l.queues[5], hit_count, l.pipe.cap);
    }
    ProfScope ps(ctx, s, K_SHADE, 0);
    const int walk_i = prm.integrator == PTR_LI_VOLPATH && !ctx->scene_all_opaque ? 1 : 0;
    ShadeArgs a = {ctx->sc, l.pipe, prm, hit_count, 0, l.queues[5], SssChains{nullptr, 0, nullptr, nullptr}, walk_i, ctx->scene_max_lobes, ctx->ctrs};
    if (ctx->scene_has_sss) { /* everything but the BSSRDF tails, then the tails as a dense list (rt_shade.h) */
        rt_launch_shade_defer(stride_blocks, s, a);
        const unsigned tail_blocks = clampu(nblocks(n_upper) / 4 + 1, 32, 2048);
        if (l.sss_planes && ctx->sss_rounds > 0) {
            enqueue_probe_chains(ctx, l, a, n_upper, cslot, tail_blocks);
            if (a.chains.exit.planes) rt_launch_shade_tail(tail_blocks, s, a); /* settled chains: no traversal; the rest stays deferred */
        }
        rt_launch_shade_sss(tail_blocks, s, a);
    } else rt_launch_shade(stride_blocks, s, a);
    hipLaunchKernelGGL(k_route, dim3(clampu(nblocks((n_upper + ROUTE_CHUNK - 1) / ROUTE_CHUNK), 32, 2048)), dim3(TPB), 0, s, (const int *)hit_count,
                       (const int *)l.queues[5], (const int *)l.pipe.route, q_next, l.queues[2], l.queues[3], l.queues[4], cnt, 1);
    if (a.chains.exit.planes) { /* the queues of the exit-point records, from the flags the tail kernel left per chain */
        int *scnt = l.sss_counts + SSS_COUNTER_STRIDE * cslot;
        const int upper = n_upper < l.sss_cap ? n_upper : l.sss_cap;
        hipLaunchKernelGGL(k_route, dim3(clampu(nblocks((upper + ROUTE_CHUNK - 1) / ROUTE_CHUNK), 32, 2048)), dim3(TPB), 0, s, (const int *)(scnt + 1),
                           (const int *)nullptr, (const int *)l.sss_route, l.sss_q[0] /* (bit 0 is never set) */, l.sss_q[0], l.sss_q[1], l.sss_q[2], scnt + SSS_EXIT_COUNTS, 0);
    }
}
/* the exit-point records of the bounce summed into L (after enqueue_resolve: their term follows the vertex's own) */
static void enqueue_resolve2(PtrContext *ctx, Lane &l, int n_upper, int cslot, int integrator) {
    int *scnt = l.sss_counts + SSS_COUNTER_STRIDE * cslot;
    const int upper = n_upper < l.sss_cap ? n_upper : l.sss_cap;
    ProfScope ps(ctx, l.stream, K_RESOLVE, 0);
    const SssChains ch = {l.sss_planes, l.sss_cap, scnt, scnt + 1, ExitNee{l.sss_nee2, l.sss_route}};
    hipLaunchKernelGGL(k_nee_resolve2, dim3(clampu(nblocks(upper), 32, 2048)), dim3(TPB), 0, l.stream, ctx->sc, l.pipe, ch, (const int *)(scnt + SSS_EXIT_COUNTS + 1),
                       (const int *)l.sss_q[0], ctx->ctrs, 0);
}
/* direct light of the bounce when a shadow / MIS ray may pass through primitives (Li_VolPath with
 * medium boundaries or emitters): the walks of VisibilityTester::Tr / IntersectTr run in-kernel */
static void enqueue_nee_walks(PtrContext *ctx, Lane &l, int n_upper, int cslot) {
    hipStream_t s = l.stream;
    int *cnt = l.counts + 4 * cslot;
    const unsigned stride_blocks = clampu(nblocks(n_upper), 32, 2048);
    {
        ProfScope ps(ctx, s, K_SHADOW, 0);
        rt_launch_shadow_walk(stride_blocks, s, ctx->sc, l.pipe, (const int *)(cnt + 2), 0, (const int *)l.queues[3], ctx->ctrs);
    }
    {
        ProfScope ps(ctx, s, K_MIS, 0);
        rt_launch_mis_walk(stride_blocks, s, ctx->sc, l.pipe, (const int *)(cnt + 1), 0, (const int *)l.queues[2], (const int *)l.queues[5], ctx->ctrs);
    }
}
static void enqueue_resolve(PtrContext *ctx, Lane &l, int n_upper, int cslot, bool shadow_closest = false) {
    hipStream_t s = l.stream;
    int *cnt = l.counts + 4 * cslot;
    ProfScope ps(ctx, s, K_RESOLVE, 0);
    hipLaunchKernelGGL(k_nee_resolve, dim3(clampu(nblocks(n_upper), 32, 2048)), dim3(TPB), 0, s, ctx->sc, l.pipe, (const int *)(cnt + 1), 0,
                       (const int *)l.queues[2], (const int *)l.queues[5], ctx->ctrs, shadow_closest ? 1 : 0);
}

/* PTR_DEBUG_SYNC=1: wait for the lane after a stage and report the stage a failure belongs to (a pass queues ~10 launches
 * per bounce and otherwise looks at the error status once, at its end) */
#define STAGE_CHECK(what)                                                                                         \
    do {                                                                                                          \
        hipError_t le_ = hipGetLastError();                                                                       \
        if (le_ == hipSuccess && ctx->debug_sync) le_ = hipStreamSynchronize(s);                                  \
        if (le_ != hipSuccess) return set_err(PTR_ERR_HIP, std::string("pass stage '") + (what) + "': " + hipGetErrorString(le_)); \
    } while (0)

/* Scenes with a BSSRDF material: the probe-chain and exit-point records of EVERY lane, made at the first ptr_render that needs
 * them (one chain record per sss_cap_div path slots; vertices beyond that walk their chain in the tail kernel).  All lanes at once:
 * an allocation of gigabytes stalls the device, and must not fall into a later call's passes.  If the memory is not there the
 * lanes run without records, as in round 2. */
static void ensure_sss_buffers(PtrContext *ctx) {
    if (!ctx->scene_has_sss || ctx->sss_rounds <= 0) return;
    for (Lane &l : ctx->lanes) {
        if (l.sss_planes || l.sss_cap != 0 || !l.pipe.cap) continue;
        const size_t want = ((((size_t)l.pipe.cap / ctx->sss_cap_div) >> PLANE_TILE_SHIFT) + 1) << PLANE_TILE_SHIFT;
        hipError_t e = hipMalloc((void **)&l.sss_planes, want * SSS_WORDS * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&l.sss_counts, SSS_COUNTER_INTS * sizeof(int));
        if (e == hipSuccess && ctx->sss_exit_records) {
            e = hipMalloc((void **)&l.sss_nee2, want * sizeof(NeeRec));
            if (e == hipSuccess) e = hipMalloc((void **)&l.sss_route, want * sizeof(int));
            for (int q = 0; q < 3 && e == hipSuccess; q++) e = hipMalloc((void **)&l.sss_q[q], want * sizeof(int));
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            lane_free_sss(l);
            l.sss_cap = -1; /* do not try again for these pass buffers */
        } else l.sss_cap = (int)want;
    }
}

/* one pass on a lane: nb samples (first, first+step, ...) of every pixel */
static int run_pass(PtrContext *ctx, Lane &l, int lane_index, int first, int step, int nb, int integrator) {
    PassParams prm;
    prm.seed = ctx->seed;
    prm.npix = ctx->npix_local;
    prm.pix_map = ctx->pix_map;
    prm.width = ctx->width;
    prm.height = ctx->height;
    prm.sample_first = first;
    prm.sample_step = step;
    prm.integrator = integrator;
    prm.nee_p1 = 1;
    hipStream_t s = l.stream;
    const int slots = ctx->npix_local * nb;
    /* feedback: if the pass this lane ran before has completed, look at how fast it thinned out and
     * stage bounces only while more than ~0.1 % of the paths are alive (at 64 M slots: 0.4 % -> 1654, 0.1 % -> 1707, 0.025 % -> 1689 Mrays/s) */
    if (!ctx->staged_fixed && l.staged_used > 0 && hipEventQuery(l.film_done) == hipSuccess) {
        const int *hc = l.counts_host; /* copied (async, pinned) at the end of that pass */
        int k = 0;
        while (k < l.staged_used && hc[4 * k] > l.slots_used / ctx->stage_div) k++;
        /* all staged bounces were still wide: stage one more next time */
        ctx->staged_bounces = k < l.staged_used ? k + 1 : l.staged_used + 1;
    }
    /* expected ray counts per bounce {path segments, shadow rays, MIS rays}, from the same feedback: they pick the tracer
     * of each launch (small launches go to the quad tracer); without feedback the grid-sizing bound stands in */
    long long est[MAX_ITERS][3];
    for (int k = 0; k < MAX_ITERS; k++) est[k][0] = est[k][1] = est[k][2] = -1;
    if (l.staged_used > 0 && l.slots_used > 0 && hipEventQuery(l.film_done) == hipSuccess) {
        const int *hc = l.counts_host;
        const double scale = (double)slots / (double)l.slots_used;
        for (int k = 0; k < l.staged_used && k < MAX_ITERS; k++) {
            est[k][0] = k == 0 ? slots : (long long)(hc[4 * (k - 1)] * scale);
            est[k][1] = (long long)(hc[4 * k + 2] * scale);
            est[k][2] = (long long)(hc[4 * k + 3] * scale);
        }
    }
    const size_t cbytes = (size_t)LANE_COUNTER_INTS * sizeof(int);
    HIPCHK(hipMemsetAsync(l.counts, 0, cbytes, s));
    if (l.sss_counts) HIPCHK(hipMemsetAsync(l.sss_counts, 0, SSS_COUNTER_INTS * sizeof(int), s));
    {
        ProfScope ps(ctx, s, K_RAYGEN, (uint64_t)slots);
        hipLaunchKernelGGL(k_raygen, dim3(nblocks(slots)), dim3(TPB), 0, s, ctx->sc, ctx->cam, l.pipe, prm, slots);
    }
    ctx->camera_rays += (uint64_t)slots;
    STAGE_CHECK("raygen");

    /* Staged bounces while the pass is wide, then one launch that runs the survivors to their end
     * (no bound on path length is needed: media boundaries without material do not count as
     * bounces, main.cpp:187-191).  No host round trip: every launch reads its count from the
     * counter block the previous bounce wrote; grids are sized from an estimate and stride over
     * whatever is there. */
    const int max_iters = integrator == PTR_LI_DIRECT ? 1 : integrator == PTR_LI_PATH ? 10 : integrator == PTR_LI_PATHSAMPLED ? 11 : 16;
    int staged = ctx->staged_bounces < max_iters ? ctx->staged_bounces : max_iters;
    if (ctx->count_traversal) staged = max_iters; /* counting passes: every ray goes through the tracer's counting variant, none through k_finish */
    if (staged < 1) staged = 1;
    if (staged > MAX_ITERS) staged = MAX_ITERS;
    const int *q_cur = nullptr; /* identity */
    int qi = 0;
    /* the walks of VisibilityTester::Tr / IntersectTr can only go past their first hit under
     * Li_VolPath and only if some primitive has no material (medium boundary, emitter) */
    /* ... and they go past it only at a bare medium boundary: where every primitive without a material is an emitter, each
     * walk is one closest-hit query (dev_nee.h nee_resolve_traced), served by the persistent tracer like every other ray */
    const bool nee_closest = integrator == PTR_LI_VOLPATH && !ctx->scene_all_opaque && !ctx->scene_passthrough && ctx->nee_closest_ok;
    const bool walks = integrator == PTR_LI_VOLPATH && !ctx->scene_all_opaque && !nee_closest;
    prm.nee_p1 = walks ? 1 : 0;
    int *const cursor_base = l.counts + LANE_CURSOR_BASE; /* ray-fetch cursors of the trace launches, 3 sets per bounce */
    auto cursors_of = [&](int i) { return cursor_base + (size_t)i * LANE_CURSOR_INTS; };
    auto upper_at = [&](int k) { int u = slots >> (k < 6 ? k : 6); return u < 1 ? 1 : u; };
    if (!walks && !nee_closest && ctx->merge_traces) {
        /* per bounce: shade, then ONE trace launch for the rays it made - the next path segments, the
         * shadow rays and the MIS rays - then the direct-light sum (before the next shade adds emission) */
        launch_ptrace(ctx, s, path_io(l, nullptr, nullptr, slots, cursors_of(0)), slots, false, ctx->count_traversal, K_TRACE);
        for (int k = 0; k < staged; k++) {
            int *q_next = l.queues[qi];
            int *cnt = l.counts + 4 * k;
            enqueue_shade(ctx, l, prm, k == 0 ? nullptr : l.counts + 4 * (k - 1), slots, upper_at(k), q_cur, q_next, k);
            const bool more = k + 1 < staged; /* after the last staged bounce the finish kernel traces the survivors itself */
            TraceIO p = path_io(l, q_next, more ? cnt + 0 : nullptr, 0, cursors_of(3 * k + 1));
            const int n_upper = (more ? upper_at(k + 1) : 0) + 2 * upper_at(k);
            {
                ProfScope ps(ctx, s, K_TRACE, 0);
                (use_quad(ctx, n_upper) ? rt_launch_qtrace3 : rt_launch_ptrace3)(ctx->count_traversal, trace_blocks(ctx, n_upper), s,
                                                                         ctx->sc, p, shadow_io(l, l.queues[3], cnt + 2, nullptr),
                                                                         mis_io(l, l.queues[4], cnt + 3, nullptr), ctx->ctrs);
            }
            enqueue_resolve(ctx, l, upper_at(k), k);
            STAGE_CHECK("bounce (merged traces)");
            q_cur = q_next;
            qi ^= 1;
        }
    } else {
        for (int k = 0; k < staged; k++) {
            int *q_next = l.queues[qi];
            const int *n_ptr = k == 0 ? nullptr : l.counts + 4 * (k - 1);
            int *fbc = l.counts + LANE_FB_COUNTS + 4 * k; /* {count, cursor} for the path segments, then for the MIS rays */
            launch_ptrace(ctx, s, path_io(l, q_cur, n_ptr, slots, cursors_of(3 * k)), upper_at(k), false, ctx->count_traversal, K_TRACE, est[k][0], l.fb_queue, fbc);
            STAGE_CHECK("trace");
            enqueue_shade(ctx, l, prm, n_ptr, slots, upper_at(k), q_cur, q_next, k);
            STAGE_CHECK("miss / shade / route");
            if (walks) enqueue_nee_walks(ctx, l, upper_at(k), k);
            else { /* a launch per ray kind */
                int *cnt = l.counts + 4 * k;
                /* The BSDF-sampled half of EstimateDirect asks what the ray hits first; when no primitive can emit and no
                 * medium can attenuate, all nee_resolve_traced reads from the answer is whether anything was hit: any hit will do. */
                TraceIO mio = mis_io(l, l.queues[4], cnt + 3, cursors_of(3 * k + 2));
                const bool mis_any = ctx->mis_any_hit && ctx->scene_all_opaque && !ctx->scene_has_media;
                mio.any_as_hit = mis_any ? 1 : 0;
                if (mis_any && ctx->nee_one_launch && !ctx->count_traversal) { /* (the counting pass of bench.py wants a launch per ray kind) */
                    ProfScope ps(ctx, s, K_SHADOW, 0);
                    (use_quad(ctx, 2ll * upper_at(k)) ? rt_launch_qtrace2any : rt_launch_ptrace2any)(trace_blocks(ctx, 2ll * upper_at(k)), s,
                                                                                     ctx->sc, shadow_io(l, l.queues[3], cnt + 2, cursors_of(3 * k + 1)), mio);
                } else if (exit_records(ctx, l, integrator) && !ctx->count_traversal) {
                    /* scenes with a BSSRDF material: the rays of the exit-point records ride in the same launches (any hit + any hit
                     * in one TwoAny launch; a closest-hit MIS launch gets a second launch for them) */
                    TraceIO mio2 = exit_io(l, k, true);
                    mio2.any_as_hit = mis_any ? 1 : 0;
                    {
                        ProfScope ps(ctx, s, K_SHADOW, 0);
                        rt_launch_ptrace2any(trace_blocks(ctx, 2ll * upper_at(k)), s, ctx->sc, shadow_io(l, l.queues[3], cnt + 2, cursors_of(3 * k + 1)), exit_io(l, k, false));
                    }
                    if (mis_any) {
                        ProfScope ps(ctx, s, K_MIS_ANY, 0);
                        rt_launch_ptrace2any(trace_blocks(ctx, 2ll * upper_at(k)), s, ctx->sc, mio, mio2);
                    } else {
                        launch_ptrace(ctx, s, mio, upper_at(k), false, false, K_MIS, est[k][2], l.fb_queue, fbc + 2);
                        launch_ptrace(ctx, s, mio2, upper_at(k), false, false, K_MIS);
                    }
                } else {
                launch_ptrace(ctx, s, shadow_io(l, l.queues[3], cnt + 2, cursors_of(3 * k + 1), nee_closest), upper_at(k), !nee_closest, ctx->count_traversal, K_SHADOW, est[k][1]);
                launch_ptrace(ctx, s, mio, upper_at(k), mis_any, ctx->count_traversal, mis_any ? K_MIS_ANY : K_MIS, est[k][2], l.fb_queue, fbc + 2);
                if (exit_records(ctx, l, integrator)) { /* (counting pass: a launch per ray kind) */
                    TraceIO mio2 = exit_io(l, k, true);
                    mio2.any_as_hit = mis_any ? 1 : 0;
                    launch_ptrace(ctx, s, exit_io(l, k, false), upper_at(k), true, ctx->count_traversal, K_SHADOW);
                    launch_ptrace(ctx, s, mio2, upper_at(k), mis_any, ctx->count_traversal, mis_any ? K_MIS_ANY : K_MIS);
                }
                }
                enqueue_resolve(ctx, l, upper_at(k), k, nee_closest);
                if (exit_records(ctx, l, integrator)) enqueue_resolve2(ctx, l, upper_at(k), k, integrator);
            }
            STAGE_CHECK("direct light");
            q_cur = q_next;
            qi ^= 1;
        }
    }
    if (integrator != PTR_LI_DIRECT) {
        const bool walk = integrator == PTR_LI_VOLPATH && !ctx->scene_all_opaque;
        int upper = slots >> (staged < 6 ? staged : 6);
        ProfScope ps(ctx, s, K_FINISH, 0);
        const FinishArgs a = {ctx->sc, l.pipe, prm, (const int *)(l.counts + 4 * (staged - 1)), 0, q_cur, walk ? 1 : 0, ctx->scene_max_lobes, ctx->ctrs};
        const unsigned blocks = clampu(nblocks(upper), 32, 2048);
        if (ctx->scene_has_sss) rt_launch_finish_sss(blocks, s, a);
        else rt_launch_finish(blocks, s, a);
        STAGE_CHECK("finish");
    }
    l.staged_used = staged;
    l.slots_used = slots;
    /* the film is accumulated in sample order: this pass's film kernel follows the previous one's */
    if (ctx->last_film_lane >= 0 && ctx->last_film_lane != lane_index)
        HIPCHK(hipStreamWaitEvent(s, ctx->lanes[ctx->last_film_lane].film_done, 0));
    /* ... and whatever the caller queued on its stream before this ptr_render (the event recorded at its entry).  Only the
     * film kernel touches memory the caller can see; everything before it works on the lane's own buffers and may run ahead.
     * (Waiting at the start of the pass would also wait for the previous call's passes, which were joined into that
     * stream, and serialise calls that are meant to overlap: measured 1660 -> 1470 Mrays/s.) */
    HIPCHK(hipStreamWaitEvent(s, ctx->entry, 0));
    {
        ProfScope ps(ctx, s, K_FILM, (uint64_t)ctx->npix_local);
        hipLaunchKernelGGL(k_film, dim3(nblocks(ctx->npix_local)), dim3(TPB), 0, s, l.pipe, ctx->npix_local, (const int *)ctx->pix_map, nb, ctx->we, ctx->acc,
                           ctx->samples);
    }
    HIPCHK(hipMemcpyAsync(l.counts_host, l.counts, 4 * MAX_ITERS * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(l.film_done, s));
    ctx->last_film_lane = lane_index;
    HIPCHK(hipGetLastError());
    return PTR_OK;
}

extern "C" {

int ptr_destroy(PtrContext *ctx);
int ptr_scene_upload_built(PtrContext *ctx, const PtrBuiltScene *built);
const char *ptr_last_error(void) { return g_err.c_str(); }

int ptr_create(int device, PtrContext **out) {
    if (!out) return set_err(PTR_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return set_err(PTR_ERR_HIP, "no HIP device");
    if (device < 0) HIPCHK(hipGetDevice(&device));
    if (device >= ndev) return set_err(PTR_ERR_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (prop.warpSize != 64) return set_err(PTR_ERR_HIP, "device wavefront size is not 64 (built for gfx950)");
    if (device >= PTR_MAX_DEVICES) return set_err(PTR_ERR_LIMIT, "device index beyond PTR_MAX_DEVICES");
    int *scratch_here = nullptr;
    { /* once per device, with that device current (hipSetDevice above) */
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (!g_scratch_ptr[device]) {
            int *p = nullptr;
            HIPCHK(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
            g_scratch_ptr[device] = p;
        }
        scratch_here = g_scratch_ptr[device];
    }
    PtrContext *ctx = new PtrContext();
    ctx->device = device;
    ctx->static_scratch = scratch_here;
    memset(&ctx->sc, 0, sizeof(ctx->sc));
    memset(&ctx->cam, 0, sizeof(ctx->cam));
    memset(ctx->prof_ms, 0, sizeof(ctx->prof_ms));
    memset(ctx->prof_launches, 0, sizeof(ctx->prof_launches));
    memset(ctx->prof_items, 0, sizeof(ctx->prof_items));
    bool ok = hipMalloc((void **)&ctx->ctrs, CTR_REPLICAS * CTR_STRIDE * sizeof(unsigned long long)) == hipSuccess &&
              hipMemset(ctx->ctrs, 0, CTR_REPLICAS * CTR_STRIDE * sizeof(unsigned long long)) == hipSuccess &&
              hipMalloc((void **)&ctx->api_fetch, API_FETCH_SLOTS * sizeof(int)) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->entry, hipEventDisableTiming) == hipSuccess;
    /* The lane streams must share ONE priority (plain hipStreamCreateWithFlags).  The film kernels of consecutive passes are
     * chained across lanes by events, in sample order (run_pass: hipStreamWaitEvent on the previous lane's film_done), and a
     * tracer launch is a persistent grid that fills every CU (persist_blocks = CUs x PT_WAVES resident blocks) until its queue
     * is drained.  With mixed priorities (round 2: hipStreamCreateWithPriority, alternating highest / lowest) the hardware
     * scheduler keeps handing CUs to the high-priority lanes' persistent grids; the pass that owns the NEXT film_done event
     * in the chain sits on a low-priority lane, gets no CU while any high-priority grid has rays left, and every
     * high-priority pass in turn ends at a film kernel that waits for that event: the six lanes multiplexed on the
     * GPU's four hardware queues made no progress for minutes.  At equal priority a launch waits only for launches queued
     * before it, each of which drains (every wave of a persistent grid reaches its exit, dev_trace.h), and the chain
     * follows the order in which ptr_render queued the passes, so it cannot close into a cycle - also with N_LANES above
     * the queue count plus the caller's stream: streams sharing a hardware queue only serialise, in that same order. */
    for (Lane &l : ctx->lanes) {
        memset(&l.pipe, 0, sizeof(l.pipe));
        ok = ok && hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&l.film_done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&l.joined, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        ptr_destroy(ctx);
        return set_err(PTR_ERR_HIP, "context setup failed (hipMalloc / stream / event)");
    }
    /* persistent grid: enough 256-thread blocks to fill every CU at the tracer's occupancy */
    ctx->n_cus = prop.multiProcessorCount;
    ctx->persist_blocks = prop.multiProcessorCount * PT_WAVES;
    g_n_cursors = PT_N_CURSORS; /* (process-wide, like the other tuning hooks of the trace launches: the last ptr_create decides) */
    if (const char *e = getenv("PTR_CURSORS")) { int v = atoi(e); if (v >= 1 && v <= PT_N_CURSORS) g_n_cursors = v; }
    if (const char *e = getenv("PTR_TRACE_WAVES")) { int v = atoi(e); if (v == PT_WAVES || v == PT_WAVES_WIDE || v == PT_WAVES_LEAN) ctx->trace_waves_forced = v; }
    if (const char *e = getenv("PTR_TRACE_LEAN")) ctx->trace_lean = atoi(e) != 0;
    if (const char *e = getenv("PTR_TRACE_WIDE_FROM")) { int v = atoi(e); if (v >= 0) ctx->wide_from = v; }
    if (const char *e = getenv("PTR_STAGED")) { /* tuning experiments: fixed number of staged bounces */
        int v = atoi(e);
        if (v > 0) { ctx->staged_bounces = v; ctx->staged_fixed = true; }
    }
    if (const char *e = getenv("PTR_DEBUG_SYNC")) ctx->debug_sync = atoi(e) != 0;
    if (const char *e = getenv("PTR_MERGE")) ctx->merge_traces = atoi(e) != 0; /* tuning experiments */
    if (const char *e = getenv("PTR_TRACER")) ctx->tracer_mode = !strcmp(e, "quad") ? 1 : (!strcmp(e, "auto") ? 2 : 0);
    if (const char *e = getenv("PTR_QUAD_BELOW")) { long long v = atoll(e); if (v >= 0) ctx->quad_below = v; }
    if (const char *e = getenv("PTR_NEE_ONE")) ctx->nee_one_launch = atoi(e) != 0;
    if (const char *e = getenv("PTR_BIN_HITS")) ctx->bin_hits = atoi(e) != 0;
    if (const char *e = getenv("PTR_STAGE_DIV")) { int v = atoi(e); if (v > 0) ctx->stage_div = v; }
    if (const char *e = getenv("PTR_MIS_ANY")) ctx->mis_any_hit = atoi(e) != 0;
    if (const char *e = getenv("PTR_FAST_TOP")) ctx->fast_top = atoi(e) != 0;
    if (const char *e = getenv("PTR_NEE_CLOSEST")) ctx->nee_closest_ok = atoi(e) != 0;
    if (const char *e = getenv("PTR_SSS_CHAINS")) ctx->sss_rounds = atoi(e) != 0 ? 1 : 0;
    if (const char *e = getenv("PTR_SSS_CAP_DIV")) { int v = atoi(e); if (v >= 1) ctx->sss_cap_div = v; }
    if (const char *e = getenv("PTR_SSS_EXIT")) ctx->sss_exit_records = atoi(e) != 0;
    if (const char *e = getenv("PTR_LANES")) { int v = atoi(e); if (v >= 1 && v <= N_LANES) { ctx->lanes_used = ctx->lanes_alloc = v; ctx->lanes_forced = true; } }
    if (const char *e = getenv("PTR_SPREAD")) { /* tuning experiments */
        int v = atoi(e);
        if (v > 0) ctx->spread = v;
    }
    if (const char *e = getenv("PTR_PERSIST_BLOCKS")) { /* tuning experiments */
        int v = atoi(e);
        if (v > 0) { ctx->persist_blocks = v; ctx->persist_blocks_forced = true; }
    }
    *out = ctx;
    return PTR_OK;
}

int ptr_destroy(PtrContext *ctx) {
    if (!ctx) return PTR_OK;
    (void)hipSetDevice(ctx->device);
    (void)sync_all(ctx);
    prof_drain(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    free_scene(ctx);
    free_film(ctx);
    for (Lane &l : ctx->lanes) {
        if (l.film_done) (void)hipEventDestroy(l.film_done);
        if (l.joined) (void)hipEventDestroy(l.joined);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    if (ctx->entry) (void)hipEventDestroy(ctx->entry);
    if (ctx->ctrs) (void)hipFree(ctx->ctrs);
    if (ctx->api_fetch) (void)hipFree(ctx->api_fetch);
    if (ctx->api_fb) (void)hipFree(ctx->api_fb);
    delete ctx;
    return PTR_OK;
}

int ptr_set_stream(PtrContext *ctx, void *hip_stream) {
    if (!ctx) return set_err(PTR_ERR_INVALID, "null context");
    ctx->stream = (hipStream_t)hip_stream;
    return PTR_OK;
}

struct PtrBuiltScene {
    HostScene hs;
};
static PtrBvhBuildInfo built_info(const HostScene &hs) {
    PtrBvhBuildInfo o = {};
    o.builder = hs.bvh_builder;
    o.depth = hs.bvh_depth;
    o.build_ms = hs.bvh_build_ms;
    o.sah_boxes = hs.sah_boxes;
    o.sah_items = hs.sah_items;
    o.sah_leaves = hs.sah_leaves;
    return o;
}

int ptr_scene_build(const PtrFlatScene *scene, PtrBuiltScene **out) {
    if (!out) return set_err(PTR_ERR_INVALID, "null out pointer");
    *out = nullptr;
    PtrBuiltScene *b = new PtrBuiltScene();
    std::string err;
    int rc = ptr_flatten(scene, &b->hs, &err);
    if (rc) {
        delete b;
        return set_err(rc, err);
    }
    *out = b;
    return PTR_OK;
}
struct PtrEnvTables {
    HostEnvTables t;
};
int ptr_envmap_build(int device, const float *rgb, int width, int height, const float *scale3, PtrEnvTables **out) {
    if (!out) return set_err(PTR_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!rgb || !scale3 || width <= 0 || height <= 0 || (int64_t)width * height > (1 << 28)) return set_err(PTR_ERR_INVALID, "bad environment map");
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return set_err(PTR_ERR_HIP, "hipSetDevice failed");
    PtrEnvTables *e = new PtrEnvTables();
    std::string err;
    int rc = ptr_gpu_build_envmap(rgb, width, height, scale3, &e->t, &e->t.build_ms, &err);
    if (rc) {
        delete e;
        return set_err(rc, err);
    }
    *out = e;
    return PTR_OK;
}
int ptr_envmap_view(const PtrEnvTables *tables, PtrEnvMap *view, double *build_ms) {
    if (!tables || !view) return set_err(PTR_ERR_INVALID, "null argument");
    const HostEnvTables &t = tables->t;
    view->width = t.width; view->height = t.height; view->nu = t.nu; view->nv = t.nv;
    view->texels = t.texels.data(); view->func = t.func.data(); view->cdf = t.cdf.data();
    view->func_int = t.func_int.data(); view->marg_cdf = t.marg_cdf.data(); view->marg_int = t.marg_int;
    if (build_ms) *build_ms = t.build_ms;
    return PTR_OK;
}
int ptr_envmap_free(PtrEnvTables *tables) {
    delete tables;
    return PTR_OK;
}
int ptr_scene_build_mode(int mode) {
    if (mode != PTR_BVH_HOST_SAH && mode != PTR_BVH_DEVICE_LBVH) return set_err(PTR_ERR_INVALID, "unknown BVH build mode");
    ptr_bvh_build_mode_set(mode);
    return PTR_OK;
}
int ptr_bvh_build_info(PtrContext *ctx, PtrBvhBuildInfo *out) {
    if (!ctx || !out) return set_err(PTR_ERR_INVALID, "null argument");
    if (!ctx->has_scene) return set_err(PTR_ERR_STATE, "no scene uploaded");
    *out = ctx->bvh_build;
    return PTR_OK;
}
int ptr_scene_built_info(const PtrBuiltScene *built, PtrBvhBuildInfo *out) {
    if (!built || !out) return set_err(PTR_ERR_INVALID, "null argument");
    *out = built_info(built->hs);
    return PTR_OK;
}
int ptr_scene_built_hash(const PtrBuiltScene *built, uint64_t *out) {
    if (!built || !out) return set_err(PTR_ERR_INVALID, "null argument");
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](const void *p, size_t n) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    };
    const HostScene &hs = built->hs;
    if (!hs.nodes.empty()) mix(hs.nodes.data(), hs.nodes.size() * sizeof(hs.nodes[0]));
    if (!hs.items.empty()) mix(hs.items.data(), hs.items.size() * sizeof(hs.items[0]));
    if (!hs.top.empty()) mix(hs.top.data(), hs.top.size() * sizeof(hs.top[0]));
    if (!hs.order.empty()) mix(hs.order.data(), hs.order.size() * sizeof(hs.order[0]));
    *out = h;
    return PTR_OK;
}
int ptr_scene_built_free(PtrBuiltScene *built) {
    delete built;
    return PTR_OK;
}
int ptr_scene_upload(PtrContext *ctx, const PtrFlatScene *scene) {
    int rc = ctx_use(ctx); /* a device build (ptr_scene_build_mode) runs on this context's GPU */
    if (rc) return rc;
    PtrBuiltScene *b = nullptr;
    rc = ptr_scene_build(scene, &b);
    if (rc) return rc;
    rc = ptr_scene_upload_built(ctx, b);
    ptr_scene_built_free(b);
    return rc;
}

/* A CDF's first `size` entries as the 16-ary search tree of dev_light.h find_interval_tree, appended to `out`; returns its levels.
 * Nodes level by level, 16 floats each: 15 separators and a pad; +inf where the array has ended. */
static int cdf_tree_levels(int size) {
    int levels = 1;
    long long cover = CDF_TREE_FAN;
    while (cover < size) { cover *= CDF_TREE_FAN; levels++; }
    return levels;
}
static size_t cdf_tree_nodes(int levels) {
    size_t n = 0, w = 1;
    for (int l = 0; l < levels; l++) { n += w; w *= CDF_TREE_FAN; }
    return n;
}
static void build_cdf_tree(const float *cdf, int size, int levels, std::vector<float> &out) {
    long long B = 1;
    for (int l = 1; l < levels; l++) B *= CDF_TREE_FAN;
    size_t width = 1;
    for (int l = 0; l < levels; l++) {
        for (size_t node = 0; node < width; node++) {
            const long long s = (long long)node * B * CDF_TREE_FAN;
            for (int j = 1; j < CDF_TREE_FAN; j++) {
                const long long i = s + j * B;
                out.push_back(i < size ? cdf[i] : INFINITY);
            }
            out.push_back(INFINITY);
        }
        width *= CDF_TREE_FAN;
        B /= CDF_TREE_FAN;
    }
}

int ptr_scene_upload_built(PtrContext *ctx, const PtrBuiltScene *built) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!built) return set_err(PTR_ERR_INVALID, "null built scene");
    const HostScene &hs = built->hs;
    if ((rc = sync_all(ctx))) return rc;
    free_scene(ctx);
    DScene sc;
    memset(&sc, 0, sizeof(sc));
#define UP(field, vec)                                                      \
    if ((rc = upload(ctx, (vec).data(), (vec).size(), &sc.field))) return rc;
    UP(nodes, hs.nodes)
    UP(qnodes, hs.qnodes)
    UP(items, hs.items)
    UP(prims, hs.prims)
    UP(mats, hs.mats)
    UP(media, hs.media)
    UP(lights, hs.lights)
    UP(tris, hs.tris)
    UP(uvidx, hs.uvidx)
    UP(order, hs.order)
    UP(top, hs.top)
    UP(toy_tris, hs.toy_tris)
    UP(env_texels, hs.env_texels)
    UP(env_func, hs.env_func)
    UP(env_cdf, hs.env_cdf)
    UP(env_func_int, hs.env_func_int)
    UP(env_marg_cdf, hs.env_marg_cdf)
#undef UP
    /* the light map's CDFs once more as 16-ary search trees (dev_light.h): up to 64 MB of them, else the arrays are searched */
    if (hs.has_env && hs.env_nu > 1 && hs.env_nv > 1) {
        const int lu = cdf_tree_levels(hs.env_nu), lv = cdf_tree_levels(hs.env_nv);
        const size_t nu_nodes = cdf_tree_nodes(lu);
        if (nu_nodes * CDF_TREE_FAN * sizeof(float) * (size_t)hs.env_nv <= ((size_t)64 << 20)) {
            std::vector<float> tu, tv;
            tu.reserve(nu_nodes * CDF_TREE_FAN * (size_t)hs.env_nv);
            for (int v = 0; v < hs.env_nv; v++) build_cdf_tree(hs.env_cdf.data() + (size_t)v * (hs.env_nu + 1), hs.env_nu, lu, tu);
            build_cdf_tree(hs.env_marg_cdf.data(), hs.env_nv, lv, tv);
            if ((rc = upload(ctx, tu.data(), tu.size(), &sc.env_cdf_tree))) return rc;
            if ((rc = upload(ctx, tv.data(), tv.size(), &sc.env_marg_tree))) return rc;
            sc.env_tree_u_levels = lu;
            sc.env_tree_u_nodes = (int)nu_nodes;
            sc.env_tree_v_levels = lv;
        }
    }
    std::vector<DImage> dimg(hs.images.size());
    for (size_t i = 0; i < hs.images.size(); i++) {
        if ((rc = upload(ctx, hs.images[i].rgb.data(), hs.images[i].rgb.size(), &dimg[i].rgb))) return rc;
        dimg[i].width = hs.images[i].w;
        dimg[i].height = hs.images[i].h;
    }
    if ((rc = upload(ctx, dimg.data(), dimg.size(), &sc.images))) return rc;
    std::vector<DMesh> dmesh(hs.meshes.size());
    for (size_t i = 0; i < hs.meshes.size(); i++) {
        dmesh[i].tri_base = hs.meshes[i].tri_base;
        dmesh[i].n_tris = hs.meshes[i].n_tris;
        dmesh[i].n_uvs = hs.meshes[i].n_uvs;
        if ((rc = upload(ctx, hs.meshes[i].uv.data(), hs.meshes[i].uv.size(), &dmesh[i].uv))) return rc;
    }
    if ((rc = upload(ctx, dmesh.data(), dmesh.size(), &sc.meshes))) return rc;
    std::vector<DCloud> dcloud(hs.clouds.size());
    for (size_t i = 0; i < hs.clouds.size(); i++) {
        dcloud[i].n = hs.clouds[i].n;
        dcloud[i].radius = hs.clouds[i].radius;
        if ((rc = upload(ctx, hs.clouds[i].pos.data(), hs.clouds[i].pos.size(), &dcloud[i].pos))) return rc;
    }
    if ((rc = upload(ctx, dcloud.data(), dcloud.size(), &sc.clouds))) return rc;
    std::vector<DTable> dtab(hs.tables.size());
    for (size_t i = 0; i < hs.tables.size(); i++) {
        const HostTable &t = hs.tables[i];
        dtab[i].nRho = t.n_rho;
        dtab[i].nRadius = t.n_radius;
        if ((rc = upload(ctx, t.rho.data(), t.rho.size(), &dtab[i].rhoSamples))) return rc;
        if ((rc = upload(ctx, t.radius.data(), t.radius.size(), &dtab[i].radiusSamples))) return rc;
        if ((rc = upload(ctx, t.profile.data(), t.profile.size(), &dtab[i].profile))) return rc;
        if ((rc = upload(ctx, t.rho_eff.data(), t.rho_eff.size(), &dtab[i].rhoEff))) return rc;
        if ((rc = upload(ctx, t.profile_cdf.data(), t.profile_cdf.size(), &dtab[i].profileCDF))) return rc;
    }
    if ((rc = upload(ctx, dtab.data(), dtab.size(), &sc.tables))) return rc;
    sc.env_marg_int = hs.env_marg_int;
    sc.env_w = hs.env_w; sc.env_h = hs.env_h; sc.env_nu = hs.env_nu; sc.env_nv = hs.env_nv;
    sc.n_prims = (int)hs.prims.size();
    sc.n_lights = (int)hs.lights.size();
    sc.n_top = hs.top_ref_nodes;
    sc.top_fast = (int)hs.top.size() > hs.top_ref_nodes && ctx->fast_top ? hs.top_ref_nodes : 0;
    sc.n_top_fast = sc.top_fast ? (int)hs.top.size() - hs.top_ref_nodes : 0;
    {
        /* the tracer build: four waves per SIMD (128 registers) once the top level holds many analytic primitives (rt_trace.hip) */
        int analytic = 0;
        for (const DPrim &pr : hs.prims) analytic += pr.type != SH_MESH && pr.type != SH_CLOUD ? 1 : 0;
        sc.has_proc = ctx->trace_lean ? 0 : 1; /* (PTR_TRACE_LEAN=0: as if every scene had them) */
        bool clouds = false;
        for (const DPrim &pr : hs.prims) {
            sc.has_proc |= pr.type == SH_SPHERE_PROC || pr.type == SH_BOX_PROC || pr.type == SH_TOY ? 1 : 0;
            clouds |= pr.type == SH_CLOUD;
        }
        /* ... and the builds without the SDF shapes' code where there is no SDF shape and the top level is small: six waves per SIMD,
         * five where particle clouds bring their double-precision quadratic (configs[3]: 111.6 ms per step with five, 114.6 with six,
         * 116.7 with the SDF code; configs[1]: 119.5 / 117.0 / 124.0) */
        /* (the four-wave build carries the SDF code: a scene of a few analytic primitives and no SDF shape - spheres around a mesh, a glass
         * sphere in a medium - is better off with the five-wave build without it: configs[4] 82.6 -> 80.7 ms per step, configs[5] 212.8 -> 205.6;
         * from 32 analytic primitives on the top level is most of a ray's work and the four-wave build's wider batch and LDS copy win) */
        const bool wide = analytic >= ctx->wide_from && (sc.has_proc || analytic >= 32);
        sc.trace_waves = ctx->trace_waves_forced ? ctx->trace_waves_forced
                                                 : (wide ? PT_WAVES_WIDE : (!sc.has_proc && !clouds && analytic < ctx->wide_from ? PT_WAVES_LEAN : PT_WAVES));
        if (sc.trace_waves == PT_WAVES_LEAN && (sc.has_proc || clouds)) sc.trace_waves = PT_WAVES; /* (forced: the lean build has neither the SDF shapes' code nor, in pair mode, the particles') */
        if (!ctx->persist_blocks_forced) ctx->persist_blocks = ctx->n_cus * sc.trace_waves;
        /* Passes in flight.  Six overlap the tail of one pass's launches with the bulk of another's; where a launch lasts 15-20 ms
         * (a top level of a hundred analytic primitives) there is no tail to speak of and the passes only take each other's
         * cache: configs[2] 671 ms per step with six, 654 with four, 641 with two, 639 with one (configs[0], [1], [5]: six is
         * best; [3], [4]: three and six within 3 %). */
        if (!ctx->lanes_forced) ctx->lanes_used = analytic >= 32 && ctx->lanes_alloc > 2 ? 2 : ctx->lanes_alloc;
    }
    sc.n_nodes = (int)hs.nodes.size();
    sc.n_items = (int)hs.items.size();
    sc.view_medium = hs.view_medium;
    sc.has_env = hs.has_env;
    ctx->sc = sc;
    ctx->bvh_bytes = (int64_t)(hs.nodes.size() * sizeof(Node128) + hs.qnodes.size() * sizeof(NodeQ) + hs.items.size() * sizeof(Item48));
    ctx->bvh_build = built_info(hs);
    ctx->scene_has_media = hs.has_media;
    ctx->scene_has_sss = !hs.tables.empty();
    ctx->scene_max_lobes = hs.max_lobes;
    ctx->scene_all_opaque = true;
    ctx->scene_passthrough = false;
    for (const DPrim &pr : hs.prims) {
        if (pr.material < 0) ctx->scene_all_opaque = false;
        if (pr.material < 0 && !(pr.flags & PF_EMISSIVE)) ctx->scene_passthrough = true;
    }
    ctx->has_scene = true;
    return PTR_OK;
}

int ptr_camera_set(PtrContext *ctx, const PtrCamera *cam) {
    if (!ctx || !cam) return set_err(PTR_ERR_INVALID, "null argument");
    DCamera c;
    memcpy(c.position, cam->position, 12);
    memcpy(c.lower_left, cam->lower_left, 12);
    memcpy(c.horizontal, cam->horizontal, 12);
    memcpy(c.vertical, cam->vertical, 12);
    memcpy(c.u, cam->u, 12);
    memcpy(c.v, cam->v, 12);
    memcpy(c.w, cam->w, 12);
    c.lens_radius = cam->lens_radius;
    ctx->cam = c;
    ctx->has_camera = true;
    return PTR_OK;
}

/* bytes of pass buffers per path slot and lane: ray 24 + hit 16 + route 4 + PathRec + NeeRec + six queues + the fallback queue */
static size_t lane_bytes_per_slot() { return 24 + 16 + 4 + sizeof(PathRec) + sizeof(NeeRec) + 7 * sizeof(int); }
/* the pass buffers of every lane for nb samples per pass; on failure everything allocated so far is freed */
static hipError_t alloc_lanes(PtrContext *ctx, int npix, int nb) {
    const size_t cap = (size_t)npix * nb;
    const size_t cap_tiles = ((cap + ((size_t)1 << PLANE_TILE_SHIFT) - 1) >> PLANE_TILE_SHIFT) << PLANE_TILE_SHIFT; /* whole tiles */
    hipError_t e = hipSuccess;
#define TRY(expr) do { if (e == hipSuccess) e = (expr); } while (0)
    for (int li = 0; li < ctx->lanes_alloc; li++) {
        Lane &l = ctx->lanes[li];
        TRY(hipMalloc((void **)&l.pipe.ro, 3 * sizeof(float) * cap));
        TRY(hipMalloc((void **)&l.pipe.rd, 3 * sizeof(float) * cap));
        TRY(hipMalloc((void **)&l.pipe.hit, sizeof(int4) * cap));
        TRY(hipMalloc((void **)&l.pipe.route, sizeof(int) * cap));
        TRY(hipMalloc((void **)&l.pipe.paths, sizeof(PathRec) * cap_tiles)); /* PATH_WORDS tiled planes */
        TRY(hipMalloc((void **)&l.pipe.nee, sizeof(NeeRec) * cap_tiles));     /* NEE_WORDS tiled planes */
        for (int q = 0; q < 6; q++) TRY(hipMalloc((void **)&l.queues[q], sizeof(int) * cap));
        if (ctx->fast_top) TRY(hipMalloc((void **)&l.fb_queue, sizeof(int) * cap));
        TRY(hipMalloc((void **)&l.counts, LANE_COUNTER_INTS * sizeof(int)));
        TRY(hipHostMalloc((void **)&l.counts_host, 4 * MAX_ITERS * sizeof(int), hipHostMallocDefault));
        if (e == hipSuccess) {
            memset(l.counts_host, 0, 4 * MAX_ITERS * sizeof(int));
            l.pipe.cap = (int)cap;
        }
    }
#undef TRY
    return e;
}

int ptr_film_init(PtrContext *ctx, int width, int height, uint64_t seed) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (width <= 0 || height <= 0 || (int64_t)width * height > (1 << 28)) return set_err(PTR_ERR_INVALID, "bad film size");
    if ((rc = sync_all(ctx))) return rc;
    int npix = width * height;
    if (!ctx->has_film || npix != ctx->npix) {
        free_film(ctx);
        /* samples per pass: launches last as long as their longest ray (~2 ms on config[1]) while
         * the marginal cost is ~0.25 ns per ray, so a pass should carry tens of millions of path slots
         * (config[1], end of round 1: 32 M -> 1578, 64 M -> 1658, 128 M -> 1694 Mrays/s; 300 B of HBM per slot and lane,
         * six lanes: 64 M slots = 117 GB of the 288 GB).  That is the target, not a requirement: the pass shrinks to what
         * the device has free (PTR_PASS_SLOTS sets another target), down to one sample of every pixel per pass. */
        long long target = 1ll << 26;
        if (const char *e = getenv("PTR_PASS_SLOTS")) {
            long long v = atoll(e);
            if (v > 0) target = v;
        }
        int nb = (int)(target / npix);
        if (nb < 1) nb = 1;
        if (nb > 64) nb = 64;
        if ((size_t)npix * nb > (size_t)1 << 29) nb = (int)(((size_t)1 << 29) / npix); /* 3 * cap is indexed with an int */
        if (nb < 1) return set_err(PTR_ERR_LIMIT, "film too large for one pass");
        hipError_t e = hipMalloc((void **)&ctx->we, 3 * sizeof(float) * npix);
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->acc, sizeof(double) * npix);
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->samples, sizeof(long long) * npix);
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->rgb8, 3 * (size_t)npix);
        if (e != hipSuccess) {
            free_film(ctx);
            return set_err(PTR_ERR_HIP, std::string("film accumulators: ") + hipGetErrorString(e));
        }
        /* start from what fits in 90 % of the free memory, then halve on failure */
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double budget = 0.9 * (double)free_b;
            while (nb > 1 && (double)npix * nb * (double)lane_bytes_per_slot() * ctx->lanes_alloc > budget) nb = (nb + 1) / 2;
        }
        for (;;) {
            e = alloc_lanes(ctx, npix, nb);
            if (e == hipSuccess) break;
            (void)hipGetLastError(); /* clear the sticky out-of-memory status */
            void *keep[] = {ctx->we, ctx->acc, ctx->samples, ctx->rgb8};
            ctx->we = nullptr; ctx->acc = nullptr; ctx->samples = nullptr; ctx->rgb8 = nullptr;
            free_film(ctx); /* the half-built lanes */
            ctx->we = (float *)keep[0]; ctx->acc = (double *)keep[1]; ctx->samples = (long long *)keep[2]; ctx->rgb8 = (unsigned char *)keep[3];
            if (nb == 1 || e != hipErrorOutOfMemory) {
                free_film(ctx);
                return set_err(PTR_ERR_HIP, std::string("pass buffers (") + std::to_string((size_t)npix * nb) + " slots): " + hipGetErrorString(e));
            }
            nb = nb / 2;
        }
        ctx->nb_max = nb;
    }
    ctx->width = width;
    ctx->height = height;
    ctx->npix = npix;
    if (ctx->pix_map) (void)hipFree(ctx->pix_map); /* a new film belongs to this context whole until ptr_film_tiles says otherwise */
    ctx->pix_map = nullptr;
    ctx->npix_local = npix;
    ctx->tile_offset = 0;
    ctx->tile_stride = 1;
    ctx->seed = seed;
    ctx->sample_cursor = 0;
    ctx->last_film_lane = -1;
    HIPCHK(hipMemset(ctx->we, 0, 3 * sizeof(float) * npix));
    HIPCHK(hipMemset(ctx->acc, 0, sizeof(double) * npix));
    HIPCHK(hipMemset(ctx->samples, 0, sizeof(long long) * npix));
    HIPCHK(hipDeviceSynchronize());
    ctx->has_film = true;
    return PTR_OK;
}

int ptr_film_shard(PtrContext *ctx, int sample_offset, int sample_stride) {
    if (!ctx) return set_err(PTR_ERR_INVALID, "null context");
    if (sample_stride < 1 || sample_offset < 0 || sample_offset >= sample_stride) return set_err(PTR_ERR_INVALID, "bad shard (offset, stride)");
    ctx->shard_offset = sample_offset;
    ctx->shard_stride = sample_stride;
    return PTR_OK;
}

int ptr_film_tiles(PtrContext *ctx, int tile_offset, int tile_stride) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_film) return set_err(PTR_ERR_STATE, "ptr_film_tiles needs a film");
    if (tile_stride < 1 || tile_offset < 0 || tile_offset >= tile_stride) return set_err(PTR_ERR_INVALID, "bad tile shard (offset, stride)");
    if ((rc = sync_all(ctx))) return rc;
    if (ctx->pix_map) (void)hipFree(ctx->pix_map);
    ctx->pix_map = nullptr;
    ctx->npix_local = ctx->npix;
    ctx->tile_offset = tile_offset;
    ctx->tile_stride = tile_stride;
    if (tile_stride == 1) return PTR_OK;
    const int T = PTR_TILE, tx = (ctx->width + T - 1) / T, ty = (ctx->height + T - 1) / T;
    std::vector<int> map;
    map.reserve((size_t)ctx->npix / tile_stride + T * T);
    for (int t = tile_offset; t < tx * ty; t += tile_stride) {
        const int x0 = (t % tx) * T, y0 = (t / tx) * T;
        for (int y = y0; y < y0 + T && y < ctx->height; y++)
            for (int x = x0; x < x0 + T && x < ctx->width; x++) map.push_back(y * ctx->width + x);
    }

    ctx->npix_local = (int)map.size();
    if (!map.empty()) {
        HIPCHK(hipMalloc((void **)&ctx->pix_map, map.size() * sizeof(int)));
        HIPCHK(hipMemcpy(ctx->pix_map, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    return PTR_OK;
}

int ptr_render(PtrContext *ctx, int ns, int integrator) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_scene || !ctx->has_camera || !ctx->has_film)
        return set_err(PTR_ERR_STATE, "ptr_render needs ptr_scene_upload, ptr_camera_set and ptr_film_init first");
    if (ns < 0) return set_err(PTR_ERR_INVALID, "negative sample count");
    if (integrator < PTR_LI_DIRECT || integrator > PTR_LI_VOLPATH) return set_err(PTR_ERR_INVALID, "unknown integrator");
    /* work the caller queued on its stream before this call (a film restore, an all-reduce of the accumulators, a
     * kernel of its own on ptr_film_device_ptrs) happens before a pass touches the film: the lanes are separate
     * non-blocking streams, so their film kernels are made to wait for this point of the caller's stream */
    HIPCHK(hipEventRecord(ctx->entry, ctx->stream));
    int begin = ctx->sample_cursor, end = begin + ns;
    ctx->sample_cursor = end;
    /* first sample index >= begin owned by this shard */
    int stride = ctx->shard_stride;
    int first = begin + ((ctx->shard_offset - begin) % stride + stride) % stride;
    bool used[N_LANES] = {};
    if (ctx->npix_local == 0) return PTR_OK; /* a tile shard without tiles */
    ensure_sss_buffers(ctx);
    /* samples per pass: what the pass buffers hold of this context's pixels */
    long long nb_cap = (long long)ctx->lanes[0].pipe.cap / ctx->npix_local;
    if (nb_cap > 64) nb_cap = 64;
    if (nb_cap < 1) nb_cap = 1;
    while (first < end) {
        int remaining = (end - first + stride - 1) / stride;
        /* spread the samples of this call over the lanes before filling a lane's batch */
        int nb = (remaining + ctx->spread - 1) / ctx->spread;
        if (nb > (int)nb_cap) nb = (int)nb_cap;
        if (nb < 1) nb = 1;
        int li = ctx->next_lane++ % ctx->lanes_used;
        rc = run_pass(ctx, ctx->lanes[li], li, first, stride, nb, integrator);
        if (rc) return rc;
        used[li] = true;
        first += nb * stride;
    }
    /* work queued on the caller's stream after this call sees the finished film */
    for (int li = 0; li < N_LANES; li++)
        if (used[li]) {
            HIPCHK(hipEventRecord(ctx->lanes[li].joined, ctx->lanes[li].stream));
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->lanes[li].joined, 0));
        }
    return PTR_OK;
}

int ptr_sync(PtrContext *ctx) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if ((rc = sync_all(ctx))) return rc;
    HIPCHK(hipGetLastError());
    return PTR_OK;
}

int ptr_film_read(PtrContext *ctx, float *we_rgb, double *acc_weight, int64_t *samples) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_film) return set_err(PTR_ERR_STATE, "no film");
    if ((rc = sync_all(ctx))) return rc;
    if (we_rgb) HIPCHK(hipMemcpy(we_rgb, ctx->we, 3 * sizeof(float) * ctx->npix, hipMemcpyDeviceToHost));
    if (acc_weight) HIPCHK(hipMemcpy(acc_weight, ctx->acc, sizeof(double) * ctx->npix, hipMemcpyDeviceToHost));
    if (samples) HIPCHK(hipMemcpy(samples, ctx->samples, sizeof(int64_t) * ctx->npix, hipMemcpyDeviceToHost));
    return PTR_OK;
}

int ptr_film_write(PtrContext *ctx, const float *we_rgb, const double *acc_weight, const int64_t *samples, int64_t next_sample) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_film) return set_err(PTR_ERR_STATE, "no film");
    if (!we_rgb || !acc_weight || !samples) return set_err(PTR_ERR_INVALID, "null accumulator array");
    if (next_sample < 0) next_sample = samples[0];
    if (next_sample < 0 || next_sample > 0x7fffffff) return set_err(PTR_ERR_INVALID, "sample index out of range");
    if ((rc = sync_all(ctx))) return rc;
    HIPCHK(hipMemcpy(ctx->we, we_rgb, 3 * sizeof(float) * ctx->npix, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->acc, acc_weight, sizeof(double) * ctx->npix, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->samples, samples, sizeof(int64_t) * ctx->npix, hipMemcpyHostToDevice));
    ctx->sample_cursor = (int)next_sample;
    return PTR_OK;
}

int ptr_film_rgb8(PtrContext *ctx, float exposure, int algorithm, unsigned char *rgb) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_film) return set_err(PTR_ERR_STATE, "no film");
    if (!rgb) return set_err(PTR_ERR_INVALID, "null output");
    if (algorithm < PTR_TONEMAP_REINHARD || algorithm > PTR_TONEMAP_NAUGHTYDOG) return set_err(PTR_ERR_INVALID, "unknown tone mapping algorithm");
    if ((rc = sync_all(ctx))) return rc;
    unsigned char *d = ctx->rgb8; /* allocated with the film */
    hipLaunchKernelGGL(k_tonemap, dim3(nblocks(ctx->npix)), dim3(TPB), 0, ctx->stream, (const float *)ctx->we, (const double *)ctx->acc, ctx->npix, exposure,
                       algorithm, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(rgb, d, 3 * (size_t)ctx->npix, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PTR_OK;
}

int ptr_film_device_ptrs(PtrContext *ctx, void **we_rgb, void **acc_weight, void **samples) {
    if (!ctx) return set_err(PTR_ERR_INVALID, "null context");
    if (!ctx->has_film) return set_err(PTR_ERR_STATE, "no film");
    if (we_rgb) *we_rgb = ctx->we;
    if (acc_weight) *acc_weight = ctx->acc;
    if (samples) *samples = ctx->samples;
    return PTR_OK;
}

/* entries (of 64 pixels per tile) in the packed film of one of `stride` tile shards: the same for every shard, sized for the largest */
static long long packed_tiles(int width, int height, int stride) {
    const long long tiles = (long long)((width + PTR_TILE - 1) / PTR_TILE) * ((height + PTR_TILE - 1) / PTR_TILE);
    return (tiles + stride - 1) / stride;
}
int ptr_film_tiles_packed_pixels(PtrContext *ctx, int tile_stride, int64_t *out) {
    if (!ctx || !out) return set_err(PTR_ERR_INVALID, "null argument");
    if (!ctx->has_film) return set_err(PTR_ERR_STATE, "no film");
    if (tile_stride < 1) return set_err(PTR_ERR_INVALID, "bad tile stride");
    *out = packed_tiles(ctx->width, ctx->height, tile_stride) * PTR_TILE * PTR_TILE;
    return PTR_OK;
}
int ptr_film_tiles_pack(PtrContext *ctx, float *d_we, double *d_acc, int64_t *d_samples) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_film) return set_err(PTR_ERR_STATE, "no film");
    if (!d_we || !d_acc || !d_samples) return set_err(PTR_ERR_INVALID, "null packed buffer");
    const long long n = packed_tiles(ctx->width, ctx->height, ctx->tile_stride) * PTR_TILE * PTR_TILE;
    hipLaunchKernelGGL(k_film_pack_tiles, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, ctx->stream, (const float *)ctx->we, (const double *)ctx->acc,
                       (const long long *)ctx->samples, ctx->width, ctx->height, ctx->tile_offset, ctx->tile_stride, n, d_we, d_acc, (long long *)d_samples);
    HIPCHK(hipGetLastError());
    return PTR_OK;
}
int ptr_film_tiles_unpack(PtrContext *ctx, int tile_offset, int tile_stride, const float *d_we, const double *d_acc, const int64_t *d_samples,
                          float *d_out_we, double *d_out_acc, int64_t *d_out_samples) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_film) return set_err(PTR_ERR_STATE, "no film");
    if (tile_stride < 1 || tile_offset < 0 || tile_offset >= tile_stride) return set_err(PTR_ERR_INVALID, "bad tile shard (offset, stride)");
    if (!d_we || !d_acc || !d_samples || !d_out_we || !d_out_acc || !d_out_samples) return set_err(PTR_ERR_INVALID, "null buffer");
    const long long n = packed_tiles(ctx->width, ctx->height, tile_stride) * PTR_TILE * PTR_TILE;
    hipLaunchKernelGGL(k_film_unpack_tiles, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, ctx->stream, d_we, d_acc, (const long long *)d_samples,
                       ctx->width, ctx->height, tile_offset, tile_stride, n, d_out_we, d_out_acc, (long long *)d_out_samples);
    HIPCHK(hipGetLastError());
    return PTR_OK;
}
int ptr_film_fill_samples(PtrContext *ctx, int64_t *d_out_samples, int64_t value) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_film) return set_err(PTR_ERR_STATE, "no film");
    if (!d_out_samples) return set_err(PTR_ERR_INVALID, "null buffer");
    hipLaunchKernelGGL(k_fill_i64, dim3(nblocks(ctx->npix)), dim3(TPB), 0, ctx->stream, (long long *)d_out_samples, (long long)ctx->npix, (long long)value);
    HIPCHK(hipGetLastError());
    return PTR_OK;
}

int ptr_counters_read(PtrContext *ctx, PtrCounters *out) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!out) return set_err(PTR_ERR_INVALID, "null out pointer");
    static_assert(sizeof(PtrCounters) == C_COUNT * sizeof(uint64_t), "counter layout");
    if ((rc = sync_all(ctx))) return rc;
    unsigned long long copies[CTR_REPLICAS * CTR_STRIDE];
    HIPCHK(hipMemcpy(copies, ctx->ctrs, sizeof(copies), hipMemcpyDeviceToHost));
    uint64_t *sum = reinterpret_cast<uint64_t *>(out);
    for (int c = 0; c < C_COUNT; c++) {
        sum[c] = 0;
        for (int r = 0; r < CTR_REPLICAS; r++) sum[c] += copies[r * CTR_STRIDE + c];
    }
    out->camera_rays = ctx->camera_rays;
    return PTR_OK;
}

int ptr_pipeline_stats_read(PtrContext *ctx, PtrPipelineStats *out) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!out) return set_err(PTR_ERR_INVALID, "null out pointer");
    static_assert(sizeof(PtrPipelineStats) == (CX_END - C_COUNT) * sizeof(uint64_t), "statistics layout");
    if ((rc = sync_all(ctx))) return rc;
    unsigned long long copies[CTR_REPLICAS * CTR_STRIDE];
    HIPCHK(hipMemcpy(copies, ctx->ctrs, sizeof(copies), hipMemcpyDeviceToHost));
    uint64_t *sum = reinterpret_cast<uint64_t *>(out);
    for (int c = C_COUNT; c < CX_END; c++) {
        sum[c - C_COUNT] = 0;
        for (int r = 0; r < CTR_REPLICAS; r++) sum[c - C_COUNT] += copies[r * CTR_STRIDE + c];
    }
    return PTR_OK;
}

int ptr_counters_reset(PtrContext *ctx) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if ((rc = sync_all(ctx))) return rc;
    HIPCHK(hipMemset(ctx->ctrs, 0, CTR_REPLICAS * CTR_STRIDE * sizeof(unsigned long long)));
    HIPCHK(hipDeviceSynchronize());
    ctx->camera_rays = 0;
    return PTR_OK;
}

int ptr_generate_rays(PtrContext *ctx, int sample, float *d_o, float *d_d, float *d_weight) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_camera || !ctx->has_film) return set_err(PTR_ERR_STATE, "ptr_generate_rays needs a camera and a film");
    if (!d_o || !d_d) return set_err(PTR_ERR_INVALID, "null ray buffers");
    PassParams prm;
    memset(&prm, 0, sizeof(prm));
    prm.seed = ctx->seed;
    prm.npix = ctx->npix;
    prm.width = ctx->width;
    prm.height = ctx->height;
    ProfScope ps(ctx, ctx->stream, K_RAYGEN, (uint64_t)ctx->npix);
    hipLaunchKernelGGL(k_raygen_only, dim3(nblocks(ctx->npix)), dim3(TPB), 0, ctx->stream, ctx->cam, prm, sample, d_o, d_d, d_weight);
    HIPCHK(hipGetLastError());
    return PTR_OK;
}

int ptr_trace_closest(PtrContext *ctx, int n, const float *d_o, const float *d_d, const float *d_tmax, int32_t *d_prim, int32_t *d_face,
                      float *d_t, float *d_b0, float *d_b1) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_scene) return set_err(PTR_ERR_STATE, "no scene uploaded");
    if (n < 0 || (n > 0 && (!d_o || !d_d || !d_prim || !d_face || !d_t))) return set_err(PTR_ERR_INVALID, "bad ray batch");
    if (n == 0) return PTR_OK;
    return api_trace(ctx, soa_io(n, nullptr, n, d_o, d_d, d_tmax, d_prim, d_face, d_t, d_b0, d_b1), false);
}

int ptr_trace_closest_noalloc(PtrContext *ctx, int n, const float *d_o, const float *d_d, const float *d_tmax, int32_t *d_prim, int32_t *d_face,
                              float *d_t, float *d_b0, float *d_b1) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_scene) return set_err(PTR_ERR_STATE, "no scene uploaded");
    if (n > PTR_TRACE_NOALLOC_MAX_RAYS) return set_err(PTR_ERR_LIMIT, "ptr_trace_closest_noalloc: more than PTR_TRACE_NOALLOC_MAX_RAYS rays in one call");
    if (n < 0 || (n > 0 && (!d_o || !d_d || !d_prim || !d_face || !d_t))) return set_err(PTR_ERR_INVALID, "bad ray batch");
    if (n == 0) return PTR_OK;
    int *cursor = ctx->static_scratch; /* the device address of scratch[] on this GPU, never the symbol */
    if (!cursor) return set_err(PTR_ERR_STATE, "static scratch not resolved for this device");
    TraceIO io = soa_io(n, nullptr, n, d_o, d_d, d_tmax, d_prim, d_face, d_t, d_b0, d_b1);
    io.counter = cursor;
    /* the array keeps what the last call left in it: the cursor must start at zero */
    HIPCHK(hipMemsetAsync(cursor, 0, PTR_SCRATCH_INTS * sizeof(int), ctx->stream));
    /* no event timing here (ProfScope creates events): straight to the launch */
    const bool quad = use_quad(ctx, n);
    const unsigned blocks = trace_blocks(ctx, n);
    if (quad) rt_launch_qtrace(false, false, blocks, ctx->stream, ctx->sc, io, ctx->ctrs);
    else rt_launch_ptrace(false, false, blocks, ctx->stream, ctx->sc, io, ctx->ctrs);
    HIPCHK(hipGetLastError());
    return PTR_OK;
}

int ptr_trace_occluded(PtrContext *ctx, int n, const float *d_o, const float *d_d, const float *d_tmax, int32_t *d_occluded) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_scene) return set_err(PTR_ERR_STATE, "no scene uploaded");
    if (n < 0 || (n > 0 && (!d_o || !d_d || !d_occluded))) return set_err(PTR_ERR_INVALID, "bad ray batch");
    if (n == 0) return PTR_OK;
    return api_trace(ctx, soa_io(n, nullptr, n, d_o, d_d, d_tmax, d_occluded, nullptr, nullptr, nullptr, nullptr), true);
}

/* ---- unit hooks (rt_unit.hip) ---- */
static const int kUnitWidths[PTR_UNIT_KINDS][2] = {{3, 1}, {10, 3}, {5, 1}, {5, 1}, {8, 1}, {7, 3}, {2, 1}, {6, 4}, {1, 1}, {2, 3}, {2, 1}, {1, 3}, {2, 1}, {2, 1}};
int ptr_unit_eval(PtrContext *ctx, int prim, int kind, int n, const float *d_in, int in_width, float *d_out, int out_width) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (kind < 0 || kind >= PTR_UNIT_KINDS) return set_err(PTR_ERR_INVALID, "unknown unit kind");
    if (in_width != kUnitWidths[kind][0] || out_width != kUnitWidths[kind][1]) return set_err(PTR_ERR_INVALID, "record widths do not match the kind");
    if (n < 0 || (n > 0 && (!d_in || !d_out))) return set_err(PTR_ERR_INVALID, "bad record arrays");
    if (kind >= PTR_UNIT_ENV_SAMPLE) {
        if (!ctx->has_scene) return set_err(PTR_ERR_STATE, "this kind needs a scene");
        if (kind <= PTR_UNIT_ENV_PDF && !ctx->sc.has_env) return set_err(PTR_ERR_STATE, "the scene has no light map");
        if (kind >= PTR_UNIT_SSS_SR) {
            if (prim < 0 || prim >= ctx->sc.n_prims) return set_err(PTR_ERR_INVALID, "primitive index out of range");
            if (!ctx->scene_has_sss) return set_err(PTR_ERR_STATE, "the scene has no subsurface material");
        }
    }
    if (n == 0) return PTR_OK;
    rt_launch_unit(ctx->stream, ctx->sc, prim, kind, n, d_in, in_width, d_out, out_width);
    HIPCHK(hipGetLastError());
    return PTR_OK;
}
int ptr_unit_bsdf(PtrContext *ctx, int prim, int sample, int n, const float *d_wo, const float *d_wi_or_u, const float *d_uv, float *d_wi_out, float *d_f,
                  float *d_pdf, int32_t *d_flags) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!ctx->has_scene) return set_err(PTR_ERR_STATE, "no scene uploaded");
    if (prim < 0 || prim >= ctx->sc.n_prims) return set_err(PTR_ERR_INVALID, "primitive index out of range");
    if (n < 0 || (n > 0 && (!d_wo || !d_wi_or_u || !d_uv || !d_f || !d_pdf || (sample && (!d_wi_out || !d_flags))))) return set_err(PTR_ERR_INVALID, "bad arrays");
    if (n == 0) return PTR_OK;
    rt_launch_unit_bsdf(ctx->stream, ctx->sc, prim, sample ? 1 : 0, n, d_wo, d_wi_or_u, d_uv, d_wi_out, d_f, d_pdf, d_flags);
    HIPCHK(hipGetLastError());
    return PTR_OK;
}

/* on: bit 0 = time every launch with HIP events, bit 1 = count child boxes / leaf items tested */
int ptr_profile_enable(PtrContext *ctx, int on) {
    if (!ctx) return set_err(PTR_ERR_INVALID, "null context");
    ctx->profiling = (on & 1) != 0;
    ctx->count_traversal = (on & 2) != 0;
    return PTR_OK;
}
int ptr_profile_reset(PtrContext *ctx) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    prof_drain(ctx);
    memset(ctx->prof_ms, 0, sizeof(ctx->prof_ms));
    memset(ctx->prof_launches, 0, sizeof(ctx->prof_launches));
    memset(ctx->prof_items, 0, sizeof(ctx->prof_items));
    return PTR_OK;
}
int ptr_profile_read(PtrContext *ctx, PtrKernelProfile *out, int max_entries, int *n_out) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (!out || max_entries < 0) return set_err(PTR_ERR_INVALID, "bad profile buffer");
    prof_drain(ctx);
    int n = max_entries < PTR_PROF_KERNELS ? max_entries : PTR_PROF_KERNELS;
    for (int i = 0; i < n; i++) {
        memset(&out[i], 0, sizeof(out[i]));
        strncpy(out[i].name, kKernelNames[i], sizeof(out[i].name) - 1);
        out[i].launches = ctx->prof_launches[i];
        out[i].total_ms = ctx->prof_ms[i];
        out[i].items = ctx->prof_items[i];
    }
    if (n_out) *n_out = n;
    return PTR_OK;
}
/* diagnostic: the probe-chain counters of the pass last run on lane `lane`, SSS_COUNTER_STRIDE ints per staged bounce (layout above) */
int ptr_debug_sss_counts(PtrContext *ctx, int lane, int *out, int n_ints, int *stride) {
    int rc = ctx_use(ctx);
    if (rc) return rc;
    if (lane < 0 || lane >= N_LANES || !out) return set_err(PTR_ERR_INVALID, "bad lane");
    if ((rc = sync_all(ctx))) return rc;
    if (stride) *stride = SSS_COUNTER_STRIDE;
    if (!ctx->lanes[lane].sss_counts) return set_err(PTR_ERR_STATE, "no probe-chain buffers on this lane");
    const int n = n_ints < SSS_COUNTER_INTS ? n_ints : SSS_COUNTER_INTS;
    HIPCHK(hipMemcpy(out, ctx->lanes[lane].sss_counts, n * sizeof(int), hipMemcpyDeviceToHost));
    return PTR_OK;
}
int ptr_bvh_info(PtrContext *ctx, int64_t *n_nodes, int64_t *n_items, int64_t *bytes) {
    if (!ctx) return set_err(PTR_ERR_INVALID, "null context");
    if (!ctx->has_scene) return set_err(PTR_ERR_STATE, "no scene uploaded");
    if (n_nodes) *n_nodes = ctx->sc.n_nodes;
    if (n_items) *n_items = ctx->sc.n_items;
    if (bytes) *bytes = ctx->bvh_bytes;
    return PTR_OK;
}

} /* extern "C" */
