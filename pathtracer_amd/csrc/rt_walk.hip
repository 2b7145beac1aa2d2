/*
 * rt_walk.hip — direct-light stages for scenes where a shadow / MIS ray may pass through
 * primitives (medium boundaries, emitters under Li_VolPath): the walks are traversed in-kernel.
 */
#include "rt_common.h"
/* The walks are chains of dependent closest-hit traversals, one ray per lane: bound by memory latency, so they want
 * occupancy more than registers (197 / 220 VGPRs unconstrained = 2 waves/SIMD). */
#ifndef WALK_WAVES
#define WALK_WAVES 2
#endif

/* direct-light stages with the walks of VisibilityTester::Tr / IntersectTr (scenes with medium
 * boundaries or emitters under Li_VolPath) */
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(WALK_WAVES, WALK_WAVES))) k_shadow_walk(DScene sc, Pipe pp, const int *n_ptr, int n_host, const int *queue,
                                                     unsigned long long *ctrs) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    LaneStack ls;
    ls.lds = &s_stack[threadIdx.x];
    NeeCtr ctr;
    ctr.shadow = ctr.mis = ctr.zero = 0;
    GRID_STRIDE_BEGIN(n_ptr, n_host)
    if (i < n) {
        const int r = queue[i]; /* direct-light record = the vertex's place in the bounce's queue of shaded vertices (k_route) */
        NeeRec rec;
        nee_load(pp, r, &rec);
        nee_shadow<true>(sc, &rec, ctr, ls);
        soa_store<WORD_OF(NeeRec, LiA), WORD_OF(NeeRec, LiA) + 3>(pp.nee, pp.cap, r, &rec);
    }
    GRID_STRIDE_END
    ctr_flush(ctrs, C_SHADOW, ctr.shadow);
}

__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(WALK_WAVES, WALK_WAVES))) k_mis_walk(DScene sc, Pipe pp, const int *n_ptr, int n_host, const int *queue,
                                                  const int *q_hit, unsigned long long *ctrs) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    LaneStack ls;
    ls.lds = &s_stack[threadIdx.x];
    NeeCtr ctr;
    ctr.shadow = ctr.mis = ctr.zero = 0;
    GRID_STRIDE_BEGIN(n_ptr, n_host)
    if (i < n) {
        const int r = queue[i];
        const int slot = q_hit[r]; /* the path the record belongs to */
        NeeRec rec;
        nee_load(pp, r, &rec);
        f3 Ld = nee_mis_resolve<true>(sc, rec, ctr, ls);
        PathRec ps;
        soa_load<0, 3>(pp.paths, pp.cap, slot, &ps);
        f3 L = ld3(ps.L);
        L += Ld;
        if (rec.flags & NEE_EXTRA) L += ld3(rec.extra);
        st3(ps.L, L);
        soa_store<0, 3>(pp.paths, pp.cap, slot, &ps);
    }
    GRID_STRIDE_END
    ctr_flush(ctrs, C_MIS, ctr.mis);
    ctr_flush(ctrs, C_ZERO, ctr.zero);
}


void rt_launch_shadow_walk(unsigned blocks, hipStream_t s, const DScene &sc, const Pipe &pp, const int *n_ptr, int n_host, const int *queue,
                           unsigned long long *ctrs) {
    hipLaunchKernelGGL(k_shadow_walk, dim3(blocks), dim3(TPB), 0, s, sc, pp, n_ptr, n_host, queue, ctrs);
}
void rt_launch_mis_walk(unsigned blocks, hipStream_t s, const DScene &sc, const Pipe &pp, const int *n_ptr, int n_host, const int *queue,
                        const int *q_hit, unsigned long long *ctrs) {
    hipLaunchKernelGGL(k_mis_walk, dim3(blocks), dim3(TPB), 0, s, sc, pp, n_ptr, n_host, queue, q_hit, ctrs);
}
