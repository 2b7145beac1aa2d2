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
        const int slot = queue[i];
        NeeRec rec;
        nee_load(pp, slot, &rec);
        nee_shadow<true>(sc, &rec, ctr, ls);
        soa_store<WORD_OF(NeeRec, LiA), WORD_OF(NeeRec, LiA) + 3>(pp.nee, pp.cap, slot, &rec);
    }
    GRID_STRIDE_END
    ctr_flush(ctrs, C_SHADOW, ctr.shadow);
}

__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(WALK_WAVES, WALK_WAVES))) k_mis_walk(DScene sc, Pipe pp, const int *n_ptr, int n_host, const int *queue,
                                                  unsigned long long *ctrs) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    LaneStack ls;
    ls.lds = &s_stack[threadIdx.x];
    NeeCtr ctr;
    ctr.shadow = ctr.mis = ctr.zero = 0;
    GRID_STRIDE_BEGIN(n_ptr, n_host)
    if (i < n) {
        int slot = queue[i];
        NeeRec rec;
        nee_load(pp, slot, &rec);
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
                        unsigned long long *ctrs) {
    hipLaunchKernelGGL(k_mis_walk, dim3(blocks), dim3(TPB), 0, s, sc, pp, n_ptr, n_host, queue, ctrs);
}

/* One round of the BSSRDF probe chains (dev_bssrdf.h): the tracer has answered the current segment of every live chain;
 * each takes the hit (make_si, count / keep on the probed primitive), sets up its next segment, and the chains that go on
 * are compacted into live_out.  live_in = null: chains 0 .. n-1 (the first round).  A streaming kernel over <= n records. */
__global__ void __launch_bounds__(TPB) k_sss_advance(DScene sc, SssChains ch, const int *n_ptr, const int *live_in, int *live_out, int *n_out,
                                                     unsigned long long *ctrs) {
    unsigned probes = 0;
    GRID_STRIDE_BEGIN(n_ptr, 0)
    bool live = false;
    int j = 0;
    if (i < n) {
        j = live_in ? live_in[i] : i;
        const ChainRef c = {ch.planes, j};
        if (c.geti(SW_STATE) == SSS_LIVE) {
            probes++; /* the segment the tracer has just traced: one probe ray (bssrdf.cpp:536) */
            live = chain_advance(sc, c);
        }
    }
    queue_append(n_out, live_out, live, j);
    GRID_STRIDE_END
    ctr_flush(ctrs, C_PROBE, probes);
}
void rt_launch_sss_advance(unsigned blocks, hipStream_t s, const DScene &sc, const SssChains &ch, const int *n_ptr, const int *live_in, int *live_out,
                           int *n_out, unsigned long long *ctrs) {
    hipLaunchKernelGGL(k_sss_advance, dim3(blocks), dim3(TPB), 0, s, sc, ch, n_ptr, live_in, live_out, n_out, ctrs);
}
