/*
 * rt_trace.hip — the persistent-wavefront tracer kernels (dev_trace.h).
 */
#include "rt_common.h"

/* Aggregator::Intersect (ANY = false) / VisibilityTester::Unoccluded (ANY = true) over a queue of
 * rays: persistent wavefronts, see dev_trace.h */
/* nodes[0 .. PT_LDS_TOP) in LDS (dev_isect.h bvh4_visit_staged); PT_LDS_TOP = 0: nothing staged, no LDS taken */
#if PT_LDS_TOP > 0
#define PT_STAGED_NODES(sc)                                  \
    __shared__ float4 s_staged[PT_LDS_TOP * PT_LDS_ROW];     \
    bvh4_stage(sc, s_staged)
#else
#define PT_STAGED_NODES(sc) const float4 *s_staged = nullptr
#endif
/* top[0 .. TOP_LDS_NODES): the upper levels of the top-level tree (the reference's own, dev_scene.h TopNode), copied to LDS
 * by every persistent block before its first ray.  PT_LDS_TOPLEVEL=0 compiles the copy out. */
#ifndef PT_LDS_TOPLEVEL
#define PT_LDS_TOPLEVEL 1
#endif
#if PT_LDS_TOPLEVEL
#define PT_STAGED_TOP(sc)                                                                         \
    __shared__ float4 s_top[TOP_LDS_NODES * TOP_LDS_ROW];                                         \
    {                                                                                             \
        const int ns_ = sc.n_top < TOP_LDS_NODES ? sc.n_top : TOP_LDS_NODES;                      \
        const float4 *src_ = reinterpret_cast<const float4 *>(sc.top);                            \
        for (int i_ = threadIdx.x; i_ < ns_ * 6; i_ += TRACE_TPB) s_top[(i_ / 6) * TOP_LDS_ROW + (i_ % 6)] = src_[i_]; \
        __syncthreads();                                                                          \
    }
#else
#define PT_STAGED_TOP(sc) const float4 *s_top = nullptr
#endif
template <bool ANY, bool COUNT>
__global__ void __launch_bounds__(TRACE_TPB) __attribute__((amdgpu_waves_per_eu(PT_WAVES, PT_WAVES))) k_ptrace(DScene sc, TraceIO io, unsigned long long *ctrs) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    PT_STAGED_NODES(sc);
    PT_STAGED_TOP(sc);
    unsigned tn = 0, ti = 0;
    OneSource<ANY> src;
    src.io = io;
    persistent_trace_any<OneSource<ANY>, COUNT>(sc, src, &s_stack[threadIdx.x], s_staged, s_top, &tn, &ti);
    if (COUNT) { /* any-hit launches are counted apart: their rays end at the first hit and cost a fraction of a closest-hit ray */
        ctr_flush(ctrs, ANY ? CX_ANY_NODES : C_NODES, tn);
        ctr_flush(ctrs, ANY ? CX_ANY_ITEMS : C_PRIMS, ti);
        if (ANY && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctrs[CX_ANY_RAYS], (unsigned long long)(io.n_ptr ? *io.n_ptr : io.n));
    }
}

/* the three ray queues of one bounce - next path segments, shadow rays, MIS rays - in one launch */
template <bool COUNT>
__global__ void __launch_bounds__(TRACE_TPB) __attribute__((amdgpu_waves_per_eu(PT_WAVES, PT_WAVES)))
k_ptrace3(DScene sc, TraceIO path, TraceIO shadow, TraceIO mis, unsigned long long *ctrs) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    PT_STAGED_NODES(sc);
    PT_STAGED_TOP(sc);
    unsigned tn = 0, ti = 0;
    ThreeSources src;
    src.path = path;
    src.shadow = shadow;
    src.mis = mis;
    persistent_trace_any<ThreeSources, COUNT>(sc, src, &s_stack[threadIdx.x], s_staged, s_top, &tn, &ti);
    if (COUNT) {
        ctr_flush(ctrs, C_NODES, tn);
        ctr_flush(ctrs, C_PRIMS, ti);
    }
}

__global__ void __launch_bounds__(TRACE_TPB) __attribute__((amdgpu_waves_per_eu(PT_WAVES, PT_WAVES))) k_ptrace2any(DScene sc, TraceIO a, TraceIO b) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    PT_STAGED_NODES(sc);
    PT_STAGED_TOP(sc);
    unsigned tn = 0, ti = 0;
    TwoAny src;
    src.a = a;
    src.b = b;
    persistent_trace_any<TwoAny, false>(sc, src, &s_stack[threadIdx.x], s_staged, s_top, &tn, &ti);
}
void rt_launch_ptrace2any(unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &a, const TraceIO &b) {
    hipLaunchKernelGGL(k_ptrace2any, dim3(blocks), dim3(TRACE_TPB), 0, s, sc, a, b);
}

/* The BSSRDF probe chains (dev_bssrdf.h): the persistent tracer with a source that continues - every chain is traced segment by
 * segment to its end inside this one launch (dev_trace.h ChainSource).  chain_advance builds a surface interaction per hit, so the
 * kernel is given more registers than the plain tracer: PCHAIN_WAVES waves per SIMD. */
#ifndef PCHAIN_WAVES
#define PCHAIN_WAVES 4
#endif
__global__ void __launch_bounds__(TRACE_TPB) __attribute__((amdgpu_waves_per_eu(PCHAIN_WAVES, PCHAIN_WAVES)))
k_pchain(DScene sc, uint32_t *planes, const int *n_ptr, int *cursor, unsigned long long *ctrs) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    PT_STAGED_NODES(sc);
    PT_STAGED_TOP(sc);
    unsigned tn = 0, ti = 0, probes = 0;
    ChainSource src;
    src.planes = planes;
    src.n_ptr = n_ptr;
    src.counter = cursor;
    src.probes = &probes;
    persistent_trace<ChainSource, false>(sc, src, &s_stack[threadIdx.x], s_staged, s_top, &tn, &ti);
    ctr_flush(ctrs, C_PROBE, probes);
}
void rt_launch_pchain(unsigned blocks, hipStream_t s, const DScene &sc, uint32_t *planes, const int *n_ptr, int *cursor, unsigned long long *ctrs) {
    hipLaunchKernelGGL(k_pchain, dim3(blocks), dim3(TRACE_TPB), 0, s, sc, planes, n_ptr, cursor, ctrs);
}

void rt_launch_ptrace3(bool count, unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &path, const TraceIO &shadow, const TraceIO &mis,
                       unsigned long long *ctrs) {
    if (count) hipLaunchKernelGGL((k_ptrace3<true>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, path, shadow, mis, ctrs);
    else hipLaunchKernelGGL((k_ptrace3<false>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, path, shadow, mis, ctrs);
}

void rt_launch_ptrace(bool any, bool count, unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &io, unsigned long long *ctrs) {
    if (any && count) hipLaunchKernelGGL((k_ptrace<true, true>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs);
    else if (any) hipLaunchKernelGGL((k_ptrace<true, false>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs);
    else if (count) hipLaunchKernelGGL((k_ptrace<false, true>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs);
    else hipLaunchKernelGGL((k_ptrace<false, false>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs);
}

#ifdef PT_STATS
/* diagnostic build only (tools/gpu_tracestats.py): read and clear the phase statistics of dev_trace.h */
extern "C" int ptr_debug_trace_stats(unsigned long long *out, int n) {
    unsigned long long tmp[PS_N] = {};
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(tmp, HIP_SYMBOL(g_pt_stats), sizeof(tmp)) != hipSuccess) return 1;
    for (int i = 0; i < n && i < PS_N; i++) out[i] = tmp[i];
    unsigned long long zero[PS_N] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_pt_stats), zero, sizeof(zero)) != hipSuccess) return 1;
    return 0;
}
#endif
