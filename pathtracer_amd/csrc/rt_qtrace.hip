/*
 * rt_qtrace.hip — the quad tracer kernels (dev_qtrace.h): four lanes per ray, stack wholly in LDS.
 */
#include "rt_common.h"
#include "dev_qtrace.h"

#ifndef QT_WAVES
#define QT_WAVES 5 /* wavefronts per SIMD the quad tracer is compiled for */
#endif

template <bool ANY, bool COUNT>
__global__ void __launch_bounds__(TRACE_TPB) __attribute__((amdgpu_waves_per_eu(QT_WAVES, QT_WAVES))) k_qtrace(DScene sc, TraceIO io, unsigned long long *ctrs) {
    __shared__ int s_stack[QT_LDS_INTS];
    unsigned tn = 0, ti = 0;
    OneSource<ANY> src;
    src.io = io;
    quad_trace<OneSource<ANY>, COUNT>(sc, src, &s_stack[threadIdx.x >> 2], &tn, &ti);
    if (COUNT) {
        ctr_flush(ctrs, C_NODES, tn);
        ctr_flush(ctrs, C_PRIMS, ti);
    }
}

template <bool COUNT>
__global__ void __launch_bounds__(TRACE_TPB) __attribute__((amdgpu_waves_per_eu(QT_WAVES, QT_WAVES)))
k_qtrace3(DScene sc, TraceIO path, TraceIO shadow, TraceIO mis, unsigned long long *ctrs) {
    __shared__ int s_stack[QT_LDS_INTS];
    unsigned tn = 0, ti = 0;
    ThreeSources src;
    src.path = path;
    src.shadow = shadow;
    src.mis = mis;
    quad_trace<ThreeSources, COUNT>(sc, src, &s_stack[threadIdx.x >> 2], &tn, &ti);
    if (COUNT) {
        ctr_flush(ctrs, C_NODES, tn);
        ctr_flush(ctrs, C_PRIMS, ti);
    }
}

__global__ void __launch_bounds__(TRACE_TPB) __attribute__((amdgpu_waves_per_eu(QT_WAVES, QT_WAVES))) k_qtrace2any(DScene sc, TraceIO a, TraceIO b) {
    __shared__ int s_stack[QT_LDS_INTS];
    unsigned tn = 0, ti = 0;
    TwoAny src;
    src.a = a;
    src.b = b;
    quad_trace<TwoAny, false>(sc, src, &s_stack[threadIdx.x >> 2], &tn, &ti);
}

void rt_launch_qtrace2any(unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &a, const TraceIO &b) {
    hipLaunchKernelGGL(k_qtrace2any, dim3(blocks), dim3(TRACE_TPB), 0, s, sc, a, b);
}
void rt_launch_qtrace3(bool count, unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &path, const TraceIO &shadow, const TraceIO &mis,
                       unsigned long long *ctrs) {
    if (count) hipLaunchKernelGGL((k_qtrace3<true>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, path, shadow, mis, ctrs);
    else hipLaunchKernelGGL((k_qtrace3<false>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, path, shadow, mis, ctrs);
}
void rt_launch_qtrace(bool any, bool count, unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &io, unsigned long long *ctrs) {
    if (any) hipLaunchKernelGGL((k_qtrace<true, false>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs);
    else if (count) hipLaunchKernelGGL((k_qtrace<false, true>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs);
    else hipLaunchKernelGGL((k_qtrace<false, false>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs);
}
