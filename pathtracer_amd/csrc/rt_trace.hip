/*
 * rt_trace.hip — the persistent-wavefront tracer kernels (dev_trace.h).
 */
#include "rt_common.h"

/* Aggregator::Intersect (ANY = false) / VisibilityTester::Unoccluded (ANY = true) over a queue of
 * rays: persistent wavefronts, see dev_trace.h */
template <bool ANY, bool COUNT>
__global__ void __launch_bounds__(TRACE_TPB) __attribute__((amdgpu_waves_per_eu(PT_WAVES, PT_WAVES))) k_ptrace(DScene sc, TraceIO io, unsigned long long *ctrs) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    unsigned tn = 0, ti = 0;
    persistent_trace<ANY, COUNT>(sc, io, &s_stack[threadIdx.x], &tn, &ti);
    if (COUNT) {
        ctr_flush(ctrs, C_NODES, tn);
        ctr_flush(ctrs, C_PRIMS, ti);
    }
}


void rt_launch_ptrace(bool any, bool count, unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &io, unsigned long long *ctrs) {
    if (any) hipLaunchKernelGGL((k_ptrace<true, false>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs);
    else if (count) hipLaunchKernelGGL((k_ptrace<false, true>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs);
    else hipLaunchKernelGGL((k_ptrace<false, false>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs);
}
