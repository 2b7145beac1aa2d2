/*
 * rt_shade.h — the shade kernel: one loop iteration of the selected integrator per live path
 * (main.cpp:72-410), see dev_path.h.  Instantiated in rt_shade.hip (MODE = SSS_NONE, SSS_DEFER) and
 * rt_shade_sss.hip (MODE = SSS_INPLACE).
 */
#pragma once
#include "rt_common.h"
#ifndef SSS_WAVES
#define SSS_WAVES 2
#endif

/* Scenes with a subsurface material shade a bounce in two launches (round 2):
 *
 *   k_shade<SSS_DEFER>    every vertex, exactly as k_shade<SSS_NONE> does - until a vertex samples the transmission lobe
 *                         of a BSSRDF material.  That vertex is dropped unfinished: nothing of it is stored (the direct-light
 *                         words it wrote are rewritten identically below), its hit word gets SHADE_DEFERRED.
 *   k_shade<SSS_INPLACE>  only the deferred vertices, from the start (the RNG stream position is part of the path record, so
 *                         the replay draws the same numbers), now with the BSSRDF tail - probe segments, exit-point
 *                         direct light - traversed in place (dev_bssrdf.h).  A block takes 1024 queue entries at a time,
 *                         compacts the deferred ones into LDS and works on that dense list.
 *
 * In round 1 one kernel did both, and a wavefront in which one lane had a tail to walk took as long as that tail: configs[4]
 * shaded a vertex in 4.2 ns against 0.6 ns elsewhere.  The tail kernel is bound by traversal latency and lives on
 * occupancy: its lobe array is dynamic LDS sized by the scene's widest material (max_lobes x 16 KB instead of 80 KB),
 * which with the 24 KB traversal stack and the 4 KB list lets two blocks share a CU when no material pushes more than
 * three lobes. */
extern __shared__ float s_dyn_lobes[];
#define SHADE_DEFERRED (1 << 30) /* in the hit word of a vertex k_shade<SSS_DEFER> left to the tail kernel */
#define SSS_CHUNK 1024

template <int MODE>
__device__ __forceinline__ void shade_slot(const DScene &sc, const Pipe &pp, const PassParams &prm, int slot, int hit_prim, float *lobes, VertexCtr &ctr,
                                           const SssCtx &sss, unsigned &shaded, unsigned &nee_recs SH_STATS_PARAM) {
    const int cap = pp.cap;
    PathRec ps;
    soa_load<0, PATH_WORDS>(pp.paths, cap, slot, &ps);
#ifdef SH_STATS
    const NeeToPlanes rec = {reinterpret_cast<float *>(pp.nee), slot, prm.nee_p1 != 0, shs};
#else
    const NeeToPlanes rec = {reinterpret_cast<float *>(pp.nee), slot, prm.nee_p1 != 0};
#endif
    f3 ro(pp.ro[slot], pp.ro[cap + slot], pp.ro[2 * cap + slot]);
    f3 rd(pp.rd[slot], pp.rd[cap + slot], pp.rd[2 * cap + slot]);
    int pixel = pass_pixel(prm, slot);
    int sample = prm.sample_first + (slot / prm.npix) * prm.sample_step;
    Rng rng = make_rng(prm.seed, (uint32_t)pixel, (uint32_t)sample);
    rng.dim = (uint32_t)ps.dim;
    f3 L = ld3(ps.L), beta = ld3(ps.beta);
    int med = ps.medium, bounces = ps.bounces, pflags = ps.flags;
    const VertexCtr ctr_before = ctr;
    SH_T(rec.st, SH_LOAD);
    VertexOut vo = shade_vertex<MODE>(sc, prm.integrator, rng, ro, rd, med, hit_prim, pp.hface[slot], pp.ht[slot], L, beta, bounces, pflags, rec, ctr, &sss,
                                      lobes);
    SH_T(rec.st, SH_TAIL);
    if (MODE == SSS_DEFER && vo.defer) { /* as if this launch had never seen the vertex */
        ctr = ctr_before;
        pp.hprim[slot] = hit_prim | SHADE_DEFERRED;
        return;
    }
    const bool alive = vo.alive, want_nee = vo.want_nee;
    shaded++;
    nee_recs += want_nee ? 1 : 0;
    const bool want_shadow = want_nee && (vo.nee_flags & NEE_A), want_mis = want_nee && (vo.nee_flags & NEE_B);
    if (MODE == SSS_INPLACE && vo.has_extra) {
        if (want_nee) { /* keeps its place in L: after this vertex's own direct-light term */
            rec.puti(NEE_W(flags), vo.nee_flags | NEE_EXTRA);
            rec.put3(NEE_W(extra), vo.L_extra);
        } else L += vo.L_extra;
    }
    st3(ps.L, L);
    st3(ps.beta, beta);
    ps.medium = med;
    ps.bounces = bounces;
    ps.flags = pflags;
    ps.dim = (int)rng.dim;
    soa_store<0, WORD_OF(PathRec, weight)>(pp.paths, cap, slot, &ps);
    if (alive) {
        pp.ro[slot] = ro.x; pp.ro[cap + slot] = ro.y; pp.ro[2 * cap + slot] = ro.z;
        pp.rd[slot] = rd.x; pp.rd[cap + slot] = rd.y; pp.rd[2 * cap + slot] = rd.z;
    }
    /* where the vertex goes next is left in its (consumed) hit word; k_route turns these into the four queues.
     * Appending here cost one atomic per wavefront and queue on a single cache line: a third of the kernel. */
    pp.hprim[slot] = (alive ? ROUTE_NEXT : 0) | (want_nee ? ROUTE_NEE : 0) | (want_shadow ? ROUTE_SHADOW : 0) | (want_mis ? ROUTE_MIS : 0);
}

template <int MODE>
__global__ void __launch_bounds__(TPB)
    __attribute__((amdgpu_waves_per_eu(MODE == SSS_INPLACE ? SSS_WAVES : SHADE_WAVES, MODE == SSS_INPLACE ? SSS_WAVES : SHADE_WAVES)))
    k_shade(DScene sc, Pipe pp, PassParams prm, const int *n_ptr, int n_host, const int *queue, int *q_next, int *q_nee, int *q_shadow, int *q_mis,
            int *counts, int walk, unsigned long long *ctrs) {
    constexpr bool TAIL = MODE == SSS_INPLACE;
    __shared__ int s_stack[TAIL ? LDS_STACK * TRACE_TPB : 1];
    __shared__ float s_lobes_static[TAIL ? 1 : LOBE_LDS_WORDS]; /* 80 KB: the BSDF lobes of the block's 256 vertices (dev_bsdf.h) */
    __shared__ int s_list[TAIL ? SSS_CHUNK : 1];
    __shared__ int s_count;
    float *s_lobes = TAIL ? s_dyn_lobes : s_lobes_static;
    LaneStack ls;
    ls.lds = &s_stack[TAIL ? threadIdx.x : 0];
    NeeCtr nctr;
    nctr.shadow = nctr.mis = nctr.zero = 0;
    unsigned probes = 0;
    const SssCtx sss = {&ls, &nctr, &probes, walk != 0};
    VertexCtr ctr;
    ctr.total = ctr.miss = ctr.surf = ctr.med = ctr.light = ctr.zero = 0;
    unsigned shaded = 0, nee_recs = 0;
#ifdef SH_STATS
    ShStats shs_, *shs = &shs_;
    for (int k = 0; k < SH_N; k++) shs_.acc[k] = 0;
#endif
    if (TAIL) {
        const int n = n_ptr ? *n_ptr : n_host;
        for (int base = blockIdx.x * SSS_CHUNK; base < n; base += gridDim.x * SSS_CHUNK) {
            if (threadIdx.x == 0) s_count = 0;
            __syncthreads();
            for (int j = threadIdx.x; j < SSS_CHUNK && base + j < n; j += TPB) {
                const int slot = queue ? queue[base + j] : base + j;
                if (pp.hprim[slot] & SHADE_DEFERRED) s_list[atomicAdd(&s_count, 1)] = slot;
            }
            __syncthreads();
            const int cnt = s_count;
            for (int j = threadIdx.x; j < cnt; j += TPB) {
                const int slot = s_list[j];
                shade_slot<MODE>(sc, pp, prm, slot, pp.hprim[slot] & ~SHADE_DEFERRED, &s_lobes[threadIdx.x], ctr, sss, shaded, nee_recs SH_STATS_ARG);
            }
            __syncthreads();
        }
    } else {
        GRID_STRIDE_BEGIN(n_ptr, n_host)
#ifdef SH_STATS
        shs->last = clock64();
        if (i < n) shs->acc[SH_VERTS] += 1; /* lane 0's count: wave-rounds that had work */
#endif
        if (i < n) {
            const int slot = queue ? queue[i] : i;
            shade_slot<MODE>(sc, pp, prm, slot, pp.hprim[slot], &s_lobes[threadIdx.x], ctr, sss, shaded, nee_recs SH_STATS_ARG);
        }
#ifdef SH_STATS
        SH_T(shs, SH_STORE);
#endif
        GRID_STRIDE_END
    }
#ifdef SH_STATS
    if (__lane_id() == 0)
        for (int k = 0; k < SH_N; k++) atomicAdd(&g_sh_stats[k], shs_.acc[k]);
#endif
    ctr_flush(ctrs, C_TOTAL, ctr.total);
    ctr_flush(ctrs, C_MISS, ctr.miss);
    ctr_flush(ctrs, C_SURF, ctr.surf);
    ctr_flush(ctrs, C_MED, ctr.med);
    ctr_flush(ctrs, C_LIGHT, ctr.light);
    ctr_flush(ctrs, C_ZERO, ctr.zero);
    ctr_flush(ctrs, CX_SHADED, shaded);
    ctr_flush(ctrs, CX_NEE_RECORDS, nee_recs);
    if (TAIL) {
        ctr_flush(ctrs, C_SHADOW, nctr.shadow);
        ctr_flush(ctrs, C_MIS, nctr.mis);
        ctr_flush(ctrs, C_PROBE, probes);
    }
}

template <int MODE> static void rt_launch_shade_t(unsigned blocks, hipStream_t s, const ShadeArgs &a) {
    size_t dyn = 0;
    if (MODE == SSS_INPLACE) {
        int lobes = a.max_lobes < 1 ? 1 : (a.max_lobes > MAX_LOBES ? MAX_LOBES : a.max_lobes);
        dyn = (size_t)lobes * LOBE_WORDS * LOBE_STRIDE * sizeof(float);
        static DynLdsOnce once;
        ensure_max_dynamic_lds(once, reinterpret_cast<const void *>(&k_shade<MODE>), (int)(MAX_LOBES * LOBE_WORDS * LOBE_STRIDE * sizeof(float)));
    }
    hipLaunchKernelGGL(k_shade<MODE>, dim3(blocks), dim3(TPB), dyn, s, a.sc, a.pp, a.prm, a.n_ptr, a.n_host, a.queue, a.q_next, a.q_nee, a.q_shadow,
                       a.q_mis, a.counts, a.walk, a.ctrs);
}
