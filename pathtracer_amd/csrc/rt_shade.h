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
 *                         words it wrote are rewritten identically below), its route word gets SHADE_DEFERRED.
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
/* Round 3: between the two launches the probe chains of the deferred vertices run as rays of the persistent tracer
 * (dev_bssrdf.h "the probe chain as queued rays"):
 *
 *   k_shade<SSS_BEGIN>    the deferred vertices again, up to the probe segment: numbers them (one atomic per 1024 queue
 *                         entries), leaves chain record j with its first segment, and j in the vertex's (unused) tA word;
 *   R rounds of           k_ptrace<closest> over the live chains' segments, k_sss_advance: make_si, count / keep, next segment;
 *   k_shade<SSS_TAIL>     the tail of every vertex whose chain is settled (finished, no more hits than the record keeps): reads
 *                         the outcome instead of walking, leaves the exit point's direct light as a second record for the
 *                         bounce's shadow / MIS launches (dev_bssrdf.h ExitNee).  No traversal code, no stack in LDS;
 *   k_shade<SSS_INPLACE>  what is still deferred - a chain the rounds did not finish or with more hits than a record keeps,
 *                         a vertex beyond the record capacity - walked in place as before.
 * The chain's traversals - most of what the tail kernel used to spend its time on, at 3 waves/SIMD - run at the tracer's
 * occupancy, and a vertex's traversal count no longer serialises inside one lane. */
extern __shared__ float s_dyn_lobes[];
#define SHADE_DEFERRED (1 << 30) /* in the route word of a vertex k_shade<SSS_DEFER> left to the tail kernels */
#define SSS_CHUNK 1024

/* where the chain number of a deferred vertex waits for the tail kernels: the tA word of its direct-light record, which only
 * the tracer's shadow launch writes, later in the bounce */
#define SSS_CHAIN_WORD WORD_OF(NeeRec, tA)
/* A deferred vertex's route word: SHADE_DEFERRED, and in bits 24..27 the route bits of its own direct-light record, which
 * k_shade<SSS_DEFER> has already written */
#define SHADE_ROUTE_SHIFT 24

/* The deferred vertices from where k_shade<SSS_DEFER> left them (dev_path.h sss_continue), MODE = SSS_BEGIN / SSS_TAIL /
 * SSS_INPLACE.  Returns false if the vertex is not this launch's (SSS_TAIL: its chain is not settled). */
template <int MODE>
__device__ __forceinline__ bool tail_slot(const DScene &sc, const Pipe &pp, const PassParams &prm, int slot, int rec_at, float *lobes, VertexCtr &ctr, SssCtx &sss) {
    const int cap = pp.cap;
    const int word = pp.route[rec_at];
    const int4 h = pp.hit[slot];
    const int hit_prim = h.x, own_route = (word >> SHADE_ROUTE_SHIFT) & (ROUTE_NEE | ROUTE_SHADOW | ROUTE_MIS);
    PathRec ps;
    soa_load<0, PATH_WORDS>(pp.paths, cap, slot, &ps);
    f3 ro(pp.ro[slot], pp.ro[cap + slot], pp.ro[2 * cap + slot]);
    f3 rd(pp.rd[slot], pp.rd[cap + slot], pp.rd[2 * cap + slot]);
    Rng rng = make_rng(prm.seed, (uint32_t)pass_pixel(prm, slot), (uint32_t)(prm.sample_first + (slot / prm.npix) * prm.sample_step));
    rng.dim = (uint32_t)ps.dim;
    f3 beta = ld3(ps.beta);
    int med = ps.medium, bounces = ps.bounces, pflags = ps.flags;
    sss.slot = slot;
    const TailOut to = sss_continue<MODE>(sc, prm.integrator, rng, ro, rd, med, hit_prim, h.y, __int_as_float(h.z), beta, bounces, pflags, ctr, &sss, lobes);
    if (MODE == SSS_BEGIN) return true;
    if (!to.settled) return false;
    if (to.has_extra) { /* direct light at the exit point, resolved in this lane: it follows the vertex's own term in L */
        if (own_route & ROUTE_NEE) {
            uint32_t *fw = pp.nee + plane_at(NEE_WORDS, rec_at, NEE_W(flags));
            *fw |= NEE_EXTRA;
            float *ex = reinterpret_cast<float *>(pp.nee);
            ex[plane_at(NEE_WORDS, rec_at, NEE_W(extra))] = to.L_extra.x;
            ex[plane_at(NEE_WORDS, rec_at, NEE_W(extra) + 1)] = to.L_extra.y;
            ex[plane_at(NEE_WORDS, rec_at, NEE_W(extra) + 2)] = to.L_extra.z;
        } else {
            f3 L = ld3(ps.L);
            L += to.L_extra;
            st3(ps.L, L);
        }
    }
    st3(ps.beta, beta);
    ps.medium = med;
    ps.bounces = bounces;
    ps.flags = pflags;
    ps.dim = (int)rng.dim;
    soa_store<0, WORD_OF(PathRec, weight)>(pp.paths, cap, slot, &ps);
    if (to.alive) {
        pp.ro[slot] = ro.x; pp.ro[cap + slot] = ro.y; pp.ro[2 * cap + slot] = ro.z;
        pp.rd[slot] = rd.x; pp.rd[cap + slot] = rd.y; pp.rd[2 * cap + slot] = rd.z;
    }
    pp.route[rec_at] = (to.alive ? ROUTE_NEXT : 0) | own_route;
    return true;
}

/* INTEG: the integrator as a compile-time constant (PTR_LI_*), or -1 = whatever prm.integrator says.  shade_vertex holds the four
 * integrators side by side; a kernel that knows which one it runs drops the others' code and, with it, their share of the
 * register allocation (k_shade<SSS_NONE>: 29 spilled registers -> 0, 5 % faster, same arithmetic) */
template <int MODE, int INTEG>
__device__ __forceinline__ void shade_slot(const DScene &sc, const Pipe &pp, const PassParams &prm, int slot, int rec_at, float *lobes, VertexCtr &ctr,
                                           const SssCtx &sss, unsigned &shaded, unsigned &nee_recs SH_STATS_PARAM) {
    const int cap = pp.cap;
    const int4 h = pp.hit[slot];
    const int hit_prim = h.x;
    PathRec ps;
    soa_load<0, PATH_WORDS>(pp.paths, cap, slot, &ps);
#ifdef SH_STATS
    const NeeToPlanes rec = {reinterpret_cast<float *>(pp.nee), rec_at, prm.nee_p1 != 0, shs};
#else
    const NeeToPlanes rec = {reinterpret_cast<float *>(pp.nee), rec_at, prm.nee_p1 != 0};
#endif
    f3 ro(pp.ro[slot], pp.ro[cap + slot], pp.ro[2 * cap + slot]);
    f3 rd(pp.rd[slot], pp.rd[cap + slot], pp.rd[2 * cap + slot]);
    int pixel = pass_pixel(prm, slot);
    int sample = prm.sample_first + (slot / prm.npix) * prm.sample_step;
    Rng rng = make_rng(prm.seed, (uint32_t)pixel, (uint32_t)sample);
    rng.dim = (uint32_t)ps.dim;
    f3 L = ld3(ps.L), beta = ld3(ps.beta);
    int med = ps.medium, bounces = ps.bounces, pflags = ps.flags;
    SH_T(rec.st, SH_LOAD);
    VertexOut vo = shade_vertex<MODE>(sc, INTEG >= 0 ? INTEG : prm.integrator, rng, ro, rd, med, hit_prim, h.y, __int_as_float(h.z), L, beta, bounces, pflags, rec, ctr, &sss,
                                      lobes);
    SH_T(rec.st, SH_TAIL);
    const bool alive = vo.alive, want_nee = vo.want_nee;
    shaded++;
    nee_recs += want_nee ? 1 : 0;
    const bool want_shadow = want_nee && (vo.nee_flags & NEE_A), want_mis = want_nee && (vo.nee_flags & NEE_B);
    if (MODE == SSS_DEFER && vo.defer) {
        /* The vertex sampled the transmission of a BSSRDF material: its own part is complete and is kept - L, beta with this
         * bounce's factor, the flags, the RNG position, the direct-light record - while the incoming ray, its medium, the bounce
         * count and the hit stay as they are for the tail kernels, which continue from here (sss_continue). */
        st3(ps.L, L);
        st3(ps.beta, beta);
        ps.flags = pflags;
        ps.dim = (int)rng.dim;
        soa_store<0, WORD_OF(PathRec, weight)>(pp.paths, cap, slot, &ps);
        pp.route[rec_at] = SHADE_DEFERRED | (((want_nee ? ROUTE_NEE : 0) | (want_shadow ? ROUTE_SHADOW : 0) | (want_mis ? ROUTE_MIS : 0)) << SHADE_ROUTE_SHIFT);
        return;
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
    /* where the vertex goes next is left in the route word of its place in the queue; k_route turns these into the four queues.
     * Appending here cost one atomic per wavefront and queue on a single cache line: a third of the kernel. */
    pp.route[rec_at] = (alive ? ROUTE_NEXT : 0) | (want_nee ? ROUTE_NEE : 0) | (want_shadow ? ROUTE_SHADOW : 0) | (want_mis ? ROUTE_MIS : 0);
}

template <int MODE, int INTEG>
__global__ void __launch_bounds__(TPB)
    __attribute__((amdgpu_waves_per_eu(MODE == SSS_INPLACE ? SSS_WAVES : SHADE_WAVES, MODE == SSS_INPLACE ? SSS_WAVES : SHADE_WAVES)))
    k_shade(DScene sc, Pipe pp, PassParams prm, const int *n_ptr, int n_host, const int *queue, SssChains chains, int walk, unsigned long long *ctrs) {
    constexpr bool BEGIN = MODE == SSS_BEGIN;
    constexpr bool TAIL = MODE == SSS_INPLACE || MODE == SSS_TAIL || BEGIN; /* works on the deferred vertices of the queue, compacted per 1024 entries */
    __shared__ int s_stack[MODE == SSS_INPLACE ? LDS_STACK * TRACE_TPB : 1];
    __shared__ int s_list[TAIL ? SSS_CHUNK : 1];
    __shared__ int s_count, s_base;
    /* the BSDF lobes of the block's 256 vertices (dev_bsdf.h): dynamic LDS, 16 KB per lobe of the scene's widest material (80 KB
     * for five lobes; a scene of one- and two-lobe materials leaves room for more blocks per CU) */
    float *s_lobes = s_dyn_lobes;
    LaneStack ls;
    ls.lds = &s_stack[MODE == SSS_INPLACE ? threadIdx.x : 0];
    NeeCtr nctr;
    nctr.shadow = nctr.mis = nctr.zero = 0;
    unsigned probes = 0;
    SssCtx sss = {&ls, &nctr, &probes, walk != 0, ChainRef{nullptr, -1}, 0, chains.exit};
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
                if (pp.route[base + j] & SHADE_DEFERRED) s_list[atomicAdd(&s_count, 1)] = base + j; /* the vertex's place in the queue = its direct-light record */
            }
            __syncthreads();
            const int cnt = s_count;
            if (BEGIN) { /* number the chains of this chunk: one atomic per 1024 queue entries */
                if (threadIdx.x == 0 && cnt > 0) {
                    s_base = atomicAdd(chains.n_chains, cnt);
                    const int got = chains.cap - s_base < cnt ? chains.cap - s_base : cnt;
                    if (got > 0) atomicAdd(chains.n_live0, got);
                }
                __syncthreads();
            }
            for (int j = threadIdx.x; j < cnt; j += TPB) {
                const int at = s_list[j];
                const int slot = queue ? queue[at] : at;
                if (BEGIN) {
                    const int cj = s_base + j;
                    const bool has = cj < chains.cap;
                    pp.nee[plane_at(NEE_WORDS, at, SSS_CHAIN_WORD)] = (uint32_t)(has ? cj : -1);
                    if (has) {
                        if (chains.exit.route) chains.exit.route[cj] = 0; /* no exit-point record yet */
                        sss.chain = ChainRef{chains.planes, cj};
                        tail_slot<SSS_BEGIN>(sc, pp, prm, slot, at, &s_lobes[threadIdx.x], ctr, sss);
                    }
                } else {
                    sss.chain = ChainRef{chains.planes, chains.planes ? (int)pp.nee[plane_at(NEE_WORDS, at, SSS_CHAIN_WORD)] : -1};
                    tail_slot<MODE == SSS_TAIL ? SSS_TAIL : SSS_INPLACE>(sc, pp, prm, slot, at, &s_lobes[threadIdx.x], ctr, sss);
                }
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
            shade_slot<MODE, INTEG>(sc, pp, prm, slot, i, &s_lobes[threadIdx.x], ctr, sss, shaded, nee_recs SH_STATS_ARG);
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
    if (BEGIN) return; /* nothing of the replay is counted */
    ctr_flush(ctrs, C_TOTAL, ctr.total);
    ctr_flush(ctrs, C_MISS, ctr.miss);
    ctr_flush(ctrs, C_SURF, ctr.surf);
    ctr_flush(ctrs, C_MED, ctr.med);
    ctr_flush(ctrs, C_LIGHT, ctr.light);
    ctr_flush(ctrs, C_ZERO, ctr.zero);
    ctr_flush(ctrs, CX_SHADED, shaded);
    ctr_flush(ctrs, CX_NEE_RECORDS, nee_recs);
    if (MODE == SSS_INPLACE) {
        ctr_flush(ctrs, C_SHADOW, nctr.shadow);
        ctr_flush(ctrs, C_MIS, nctr.mis);
        ctr_flush(ctrs, C_PROBE, probes);
    }
}

template <int MODE, int INTEG> static void rt_launch_shade_ti(unsigned blocks, hipStream_t s, const ShadeArgs &a) {
    const int lobes = a.max_lobes < 1 ? 1 : (a.max_lobes > MAX_LOBES ? MAX_LOBES : a.max_lobes);
    const size_t dyn = (size_t)lobes * LOBE_WORDS * LOBE_STRIDE * sizeof(float);
    static DynLdsOnce once;
    ensure_max_dynamic_lds(once, reinterpret_cast<const void *>(&k_shade<MODE, INTEG>), (int)(MAX_LOBES * LOBE_WORDS * LOBE_STRIDE * sizeof(float)));
    hipLaunchKernelGGL((k_shade<MODE, INTEG>), dim3(blocks), dim3(TPB), dyn, s, a.sc, a.pp, a.prm, a.n_ptr, a.n_host, a.queue, a.chains, a.walk, a.ctrs);
}
template <int MODE> static void rt_launch_shade_t(unsigned blocks, hipStream_t s, const ShadeArgs &a) {
    /* the kernels that shade every vertex of a bounce get a build for the shipped integrator (main.cpp:447); the tails run the generic one */
    if constexpr (MODE == SSS_NONE || MODE == SSS_DEFER) {
        if (a.prm.integrator == PTR_LI_VOLPATH) return rt_launch_shade_ti<MODE, PTR_LI_VOLPATH>(blocks, s, a);
    }
    rt_launch_shade_ti<MODE, -1>(blocks, s, a);
}
