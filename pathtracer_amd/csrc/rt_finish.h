/*
 * rt_finish.h — the finish kernel (whole remaining paths in one launch).  Instantiated in
 * rt_finish.hip (SSS = false) and rt_finish_sss.hip (SSS = true).
 */
#pragma once
#include "rt_common.h"

/* Finishes every path of the queue inside one launch: trace -> shade -> direct light -> ... until
 * the path ends (the loop of Li_* itself, main.cpp:105-410, one path per lane).  Used once the
 * wavefront stages have thinned the pass to a few per cent of its paths: from there on a bounce
 * is worth ~1e4 rays but would still cost three launches that each last as long as their longest
 * ray.  Same device functions, same order of operations per path as the staged kernels. */
/* Like the BSSRDF tail kernel (rt_shade.h) this one traverses in place, one path per lane, and lives on occupancy: its lobe
 * array is dynamic LDS sized by the scene's widest material, FINISH_WAVES blocks per CU fit when that leaves room. */
#ifndef FINISH_WAVES
#define FINISH_WAVES 2
#endif
extern __shared__ float s_dyn_finish_lobes[];
template <bool SSS>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(FINISH_WAVES, FINISH_WAVES))) k_finish(DScene sc, Pipe pp, PassParams prm, const int *n_ptr, int n_host, const int *queue, int walk,
                                                unsigned long long *ctrs) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    float *s_lobes = s_dyn_finish_lobes;
    LaneStack ls;
    ls.lds = &s_stack[threadIdx.x];
    VertexCtr ctr;
    ctr.total = ctr.miss = ctr.surf = ctr.med = ctr.light = ctr.zero = 0;
    NeeCtr nctr;
    nctr.shadow = nctr.mis = nctr.zero = 0;
    unsigned probes = 0;
    const SssCtx sss = {&ls, &nctr, &probes, walk != 0};
    GRID_STRIDE_BEGIN(n_ptr, n_host)
    if (i < n) {
        const int slot = queue ? queue[i] : i;
        const int cap = pp.cap;
        PathRec ps;
        soa_load<0, PATH_WORDS>(pp.paths, cap, slot, &ps);
        f3 ro(pp.ro[slot], pp.ro[cap + slot], pp.ro[2 * cap + slot]);
        f3 rd(pp.rd[slot], pp.rd[cap + slot], pp.rd[2 * cap + slot]);
        Rng rng = make_rng(prm.seed, (uint32_t)pass_pixel(prm, slot), (uint32_t)(prm.sample_first + (slot / prm.npix) * prm.sample_step));
        rng.dim = (uint32_t)ps.dim;
        f3 L = ld3(ps.L), beta = ld3(ps.beta);
        int med = ps.medium, bounces = ps.bounces, pflags = ps.flags;
        NeeRec rec_local;
        NeeRec *rec = &rec_local;
        for (int guard = 0; guard < 100000; guard++) {
            float tmax = kInfinity, b0 = 0, b1 = 0;
            int hp = -1, hf = -1;
            bool hit = traverse<false, false>(sc, ro, rd, tmax, hp, hf, b0, b1, nullptr, ls);
            VertexOut vo = shade_vertex<SSS>(sc, prm.integrator, rng, ro, rd, med, hit ? hp : -1, hit ? hf : -1, tmax, L, beta, bounces, pflags, NeeToStruct{rec}, ctr, &sss, &s_lobes[threadIdx.x]);
            if (vo.want_nee) {
                f3 Ld;
                if (walk) {
                    if (rec->flags & NEE_A) nee_shadow<true>(sc, rec, nctr, ls);
                    Ld = nee_mis_resolve<true>(sc, *rec, nctr, ls);
                } else {
                    if (rec->flags & NEE_A) nee_shadow<false>(sc, rec, nctr, ls);
                    Ld = nee_mis_resolve<false>(sc, *rec, nctr, ls);
                }
                L += Ld;
            }
            if (SSS && vo.has_extra) L += vo.L_extra;
            if (!vo.alive) break;
        }
        st3(ps.L, L);
        st3(ps.beta, beta);
        ps.medium = med;
        ps.bounces = bounces;
        ps.flags = pflags;
        ps.dim = (int)rng.dim;
        soa_store<0, WORD_OF(PathRec, weight)>(pp.paths, cap, slot, &ps);
    }
    GRID_STRIDE_END
    ctr_flush(ctrs, C_TOTAL, ctr.total);
    ctr_flush(ctrs, C_MISS, ctr.miss);
    ctr_flush(ctrs, C_SURF, ctr.surf);
    ctr_flush(ctrs, C_MED, ctr.med);
    ctr_flush(ctrs, C_LIGHT, ctr.light);
    ctr_flush(ctrs, C_ZERO, ctr.zero + nctr.zero);
    ctr_flush(ctrs, C_SHADOW, nctr.shadow);
    ctr_flush(ctrs, C_MIS, nctr.mis);
    ctr_flush(ctrs, CX_FINISH_PATHS, ctr.total);
    ctr_flush(ctrs, CX_FINISH_NEE, nctr.shadow + nctr.mis);
    if (SSS) ctr_flush(ctrs, C_PROBE, probes);
}


template <bool SSS> static void rt_launch_finish_t(unsigned blocks, hipStream_t s, const FinishArgs &a) {
    const int lobes = a.max_lobes < 1 ? 1 : (a.max_lobes > MAX_LOBES ? MAX_LOBES : a.max_lobes);
    const size_t dyn = (size_t)lobes * LOBE_WORDS * LOBE_STRIDE * sizeof(float);
    static DynLdsOnce once;
    ensure_max_dynamic_lds(once, reinterpret_cast<const void *>(&k_finish<SSS>), (int)(MAX_LOBES * LOBE_WORDS * LOBE_STRIDE * sizeof(float)));
    hipLaunchKernelGGL(k_finish<SSS>, dim3(blocks), dim3(TPB), dyn, s, a.sc, a.pp, a.prm, a.n_ptr, a.n_host, a.queue, a.walk, a.ctrs);
}
