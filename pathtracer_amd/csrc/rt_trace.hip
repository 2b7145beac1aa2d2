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
/* The upper levels of the top-level tree the kernel's rays walk - the fast tree (FAST kernels; dev_trace.h, DScene::top_fast) or
 * the reference's own (dev_scene.h TopNode) - copied to LDS by every persistent block before its first ray:
 * top[root .. root + TOP_LDS_NODES).  PT_LDS_TOPLEVEL=0 compiles the copy out. */
#ifndef PT_LDS_TOPLEVEL
#define PT_LDS_TOPLEVEL 1
#endif
#if PT_LDS_TOPLEVEL
/* How many: 64 (7 KB) in the five-wave build, whose five blocks per CU leave 32 KB each beside the 24 KB stack; 128 (14 KB) in the
 * four-wave build, which has 40 KB each - all 111 inner nodes of configs[2]'s tree (round 2 had to choose between them and the
 * fourth wave: 128 nodes meant three blocks per CU). */
/* The lean build (six waves) serves scenes of a few primitives: 8 nodes (1 KB), so that six blocks of 24 KB stack fit a CU's 160 KB. */
#define PT_TOP_STAGED(waves) ((waves) >= PT_WAVES_LEAN ? 8 : ((waves) <= PT_WAVES_WIDE ? 2 * TOP_LDS_NODES : TOP_LDS_NODES))
#define PT_TOP_BATCH_OF(waves) ((waves) <= PT_WAVES_WIDE ? PT_TOP_BATCH_WIDE : PT_TOP_BATCH)
#define PT_STAGED_TOP(sc, fast, count)                                                            \
    __shared__ float4 s_top[(count) * TOP_LDS_ROW];                                               \
    {                                                                                             \
        const int nt_ = (fast) ? sc.n_top_fast : sc.n_top;                                        \
        const int ns_ = nt_ < (count) ? nt_ : (count);                                            \
        const float4 *src_ = reinterpret_cast<const float4 *>(sc.top + ((fast) ? sc.top_fast : 0)); \
        for (int i_ = threadIdx.x; i_ < ns_ * 6; i_ += TRACE_TPB) s_top[(i_ / 6) * TOP_LDS_ROW + (i_ % 6)] = src_[i_]; \
        __syncthreads();                                                                          \
    }
#else
#define PT_TOP_STAGED(waves) TOP_LDS_NODES
#define PT_STAGED_TOP(sc, fast, count) const float4 *s_top = nullptr
#endif
/* WAVES: wavefronts per SIMD the kernel is compiled for.  Five (96 registers) where a ray spends its time in the meshes; four
 * (128 registers) for scenes with many analytic primitives, whose tests - SDF marching, double-precision quadratics - spill at 96
 * (DScene::trace_waves, ptr_runtime.hip; configs[2]: 22.5 against 16.4 ms per closest-hit launch, configs[1]: 6.77 against 6.80). */
/* PROC: the scene has SDF shapes (procedural sphere / box, origami toys).  A build without their sphere tracing needs ~100 registers
 * and spills none at 96, 15-21 at 80; with it the same kernel spills 70-80 at 96 (DScene::has_proc -> trace_waves = PT_WAVES_LEAN;
 * configs[1]: 6.44 -> 6.2 ms per closest-hit launch). */
/* Pair mode (dev_trace.h): the leaf phase deals (ray, triangle) pairs to all lanes; 64 bytes of LDS per wavefront for the round's pair
 * list.  The lean build serves scenes of triangle meshes and analytic primitives only (ptr_runtime.hip picks it for scenes without SDF
 * shapes and without particle clouds); the others keep the item loop for the leaves of clouds (configs[2], 6 % of whose tracer time is
 * spent in meshes: 601-615 ms per step with pair mode in its four-wave build, 623-628 without; configs[4]: 1948 against 1832 Mrays/s).
 * PT_PAIRS: 0 none, 1 the lean build, 2 all. */
#ifndef PT_PAIRS
#define PT_PAIRS 2
#endif
#define PT_PAIRS_OF(waves) ((waves) >= PT_WAVES_LEAN ? (PT_PAIRS >= 1 ? PT_PAIRS_TRI : PT_PAIRS_OFF) : (PT_PAIRS >= 2 ? PT_PAIRS_MIXED : PT_PAIRS_OFF))
#define PT_PAIR_LIST(waves)                                        \
    __shared__ unsigned char s_pairs[PT_PAIRS_OF(waves) ? TRACE_TPB : 1]; \
    unsigned char *const pair_list = PT_PAIRS_OF(waves) ? &s_pairs[threadIdx.x & ~63u] : nullptr
template <bool ANY, bool COUNT, bool FAST, int WAVES, bool PROC>
__global__ void __launch_bounds__(TRACE_TPB) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) k_ptrace(DScene sc, TraceIO io, unsigned long long *ctrs) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    PT_STAGED_NODES(sc);
    PT_STAGED_TOP(sc, FAST, PT_TOP_STAGED(WAVES));
    PT_PAIR_LIST(WAVES);
    unsigned tn = 0, ti = 0;
    OneSource<ANY> src;
    src.io = io;
    persistent_trace_any<OneSource<ANY>, COUNT, FAST, PT_TOP_STAGED(WAVES), PT_TOP_BATCH_OF(WAVES), PROC, PT_PAIRS_OF(WAVES)>(sc, src, &s_stack[threadIdx.x], s_staged, s_top, &tn, &ti, pair_list);
    if (!ANY && !FAST && io.is_fallback && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctrs[CX_FAST_DEFERRED], (unsigned long long)(io.n_ptr ? *io.n_ptr : io.n));
    if (COUNT) { /* any-hit launches are counted apart: their rays end at the first hit and cost a fraction of a closest-hit ray */
        ctr_flush(ctrs, ANY ? CX_ANY_NODES : C_NODES, tn);
        ctr_flush(ctrs, ANY ? CX_ANY_ITEMS : C_PRIMS, ti);
        if (ANY && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctrs[CX_ANY_RAYS], (unsigned long long)(io.n_ptr ? *io.n_ptr : io.n));
    }
}

/* closest hit with barycentrics stored (the stand-alone ptr_trace_closest): the two registers they live in are the pipeline's
 * launches' to spare */
template <bool COUNT, bool FAST, int WAVES, bool PROC>
__global__ void __launch_bounds__(TRACE_TPB) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) k_ptrace_bary(DScene sc, TraceIO io, unsigned long long *ctrs) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    PT_STAGED_NODES(sc);
    PT_STAGED_TOP(sc, FAST, PT_TOP_STAGED(WAVES));
    PT_PAIR_LIST(WAVES);
    unsigned tn = 0, ti = 0;
    OneSource<false, true> src;
    src.io = io;
    persistent_trace_any<OneSource<false, true>, COUNT, FAST, PT_TOP_STAGED(WAVES), PT_TOP_BATCH_OF(WAVES), PROC, PT_PAIRS_OF(WAVES)>(sc, src, &s_stack[threadIdx.x], s_staged, s_top, &tn, &ti, pair_list);
    if (!FAST && io.is_fallback && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctrs[CX_FAST_DEFERRED], (unsigned long long)(io.n_ptr ? *io.n_ptr : io.n));
    if (COUNT) {
        ctr_flush(ctrs, C_NODES, tn);
        ctr_flush(ctrs, C_PRIMS, ti);
    }
}

/* the three ray queues of one bounce - next path segments, shadow rays, MIS rays - in one launch */
template <bool COUNT>
__global__ void __launch_bounds__(TRACE_TPB) __attribute__((amdgpu_waves_per_eu(PT_WAVES, PT_WAVES)))
k_ptrace3(DScene sc, TraceIO path, TraceIO shadow, TraceIO mis, unsigned long long *ctrs) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    PT_STAGED_NODES(sc);
    PT_STAGED_TOP(sc, false, TOP_LDS_NODES);
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

template <bool FAST, int WAVES, bool PROC>
__global__ void __launch_bounds__(TRACE_TPB) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) k_ptrace2any(DScene sc, TraceIO a, TraceIO b) {
    __shared__ int s_stack[LDS_STACK * TRACE_TPB];
    PT_STAGED_NODES(sc);
    PT_STAGED_TOP(sc, FAST, PT_TOP_STAGED(WAVES));
    PT_PAIR_LIST(WAVES);
    unsigned tn = 0, ti = 0;
    TwoAny src;
    src.a = a;
    src.b = b;
    persistent_trace_any<TwoAny, false, FAST, PT_TOP_STAGED(WAVES), PT_TOP_BATCH_OF(WAVES), PROC, PT_PAIRS_OF(WAVES)>(sc, src, &s_stack[threadIdx.x], s_staged, s_top, &tn, &ti, pair_list);
}
/* the builds a launch chooses from: four waves per SIMD (scenes with many analytic primitives), five, and - without the SDF shapes' code - five
 * (scenes with particle clouds, whose double-precision quadratic wants the registers) and six */
#define PT_PICK_BUILD(LAUNCH)                                                \
    do {                                                                     \
        if (sc.trace_waves == PT_WAVES_WIDE) LAUNCH(PT_WAVES_WIDE, true);    \
        else if (sc.trace_waves == PT_WAVES_LEAN) LAUNCH(PT_WAVES_LEAN, false); \
        else if (!sc.has_proc) LAUNCH(PT_WAVES, false);                      \
        else LAUNCH(PT_WAVES, true);                                         \
    } while (0)
void rt_launch_ptrace2any(unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &a, const TraceIO &b) {
#define PT_L2(W, P) hipLaunchKernelGGL((k_ptrace2any<true, W, P>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, a, b)
#define PT_L2N(W, P) hipLaunchKernelGGL((k_ptrace2any<false, W, P>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, a, b)
    if (sc.top_fast) PT_PICK_BUILD(PT_L2); /* any hit: every ray may walk the fast tree */
    else PT_PICK_BUILD(PT_L2N);
#undef PT_L2
#undef PT_L2N
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
    PT_STAGED_TOP(sc, false, TOP_LDS_NODES);
    __shared__ unsigned char s_pairs[PT_PAIRS ? TRACE_TPB : 1]; /* (pair mode whatever PCHAIN_WAVES is: a probe ray's time is spent in the mesh it probes) */
    unsigned char *const pair_list = PT_PAIRS ? &s_pairs[threadIdx.x & ~63u] : nullptr;
    unsigned tn = 0, ti = 0, probes = 0;
    ChainSource src;
    src.planes = planes;
    src.n_ptr = n_ptr;
    src.counter = cursor;
    src.probes = &probes;
    persistent_trace<ChainSource, false, false, TOP_LDS_NODES, PT_TOP_BATCH, true, PT_PAIRS ? PT_PAIRS_MIXED : PT_PAIRS_OFF>(sc, src, &s_stack[threadIdx.x], s_staged, s_top, &tn, &ti, pair_list);
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

/* Any-hit rays walk the fast top-level tree whenever the scene has one; closest-hit rays when the launch also brings a fallback
 * queue for the rays whose answer may need the reference's order (dev_trace.h): the caller then launches again over that queue,
 * with io.fb_queue = null. */
bool rt_ptrace_is_fast(bool any, const DScene &sc, const TraceIO &io) { return sc.top_fast != 0 && (any || (io.fb_queue && io.fb_count)); }
void rt_launch_ptrace(bool any, bool count, unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &io, unsigned long long *ctrs) {
    const bool fast = rt_ptrace_is_fast(any, sc, io);
#define PT_LAUNCH(A, C, F)                                                                                                     \
    do {                                                                                                                       \
        if (sc.trace_waves == PT_WAVES_WIDE) hipLaunchKernelGGL((k_ptrace<A, C, F, PT_WAVES_WIDE, true>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs); \
        else if (sc.trace_waves == PT_WAVES_LEAN) hipLaunchKernelGGL((k_ptrace<A, C, F, PT_WAVES_LEAN, false>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs); \
        else if (!sc.has_proc) hipLaunchKernelGGL((k_ptrace<A, C, F, PT_WAVES, false>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs); \
        else hipLaunchKernelGGL((k_ptrace<A, C, F, PT_WAVES, true>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs);       \
    } while (0)
#define PT_LAUNCH_BARY(C, F)                                                                                                   \
    do {                                                                                                                       \
        if (sc.trace_waves == PT_WAVES_WIDE) hipLaunchKernelGGL((k_ptrace_bary<C, F, PT_WAVES_WIDE, true>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs); \
        else if (sc.trace_waves == PT_WAVES_LEAN) hipLaunchKernelGGL((k_ptrace_bary<C, F, PT_WAVES_LEAN, false>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs); \
        else if (!sc.has_proc) hipLaunchKernelGGL((k_ptrace_bary<C, F, PT_WAVES, false>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs); \
        else hipLaunchKernelGGL((k_ptrace_bary<C, F, PT_WAVES, true>), dim3(blocks), dim3(TRACE_TPB), 0, s, sc, io, ctrs);     \
    } while (0)
    if (!any && !io.tile_words && (io.b0 || io.b1)) {
        if (fast) { if (count) PT_LAUNCH_BARY(true, true); else PT_LAUNCH_BARY(false, true); }
        else { if (count) PT_LAUNCH_BARY(true, false); else PT_LAUNCH_BARY(false, false); }
    } else if (fast) {
        if (any && count) PT_LAUNCH(true, true, true);
        else if (any) PT_LAUNCH(true, false, true);
        else if (count) PT_LAUNCH(false, true, true);
        else PT_LAUNCH(false, false, true);
    } else {
        if (any && count) PT_LAUNCH(true, true, false);
        else if (any) PT_LAUNCH(true, false, false);
        else if (count) PT_LAUNCH(false, true, false);
        else PT_LAUNCH(false, false, false);
    }
#undef PT_LAUNCH
#undef PT_LAUNCH_BARY
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
