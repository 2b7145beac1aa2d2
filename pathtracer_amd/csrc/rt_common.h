/*
 * rt_common.h — what the kernel translation units of the wavefront pipeline share: the per-pass
 * buffers, record planes, queue / counter helpers, and the host-side launchers each unit exports.
 * The kernels are split over several .hip files only so that they compile in parallel; device
 * functions all live in the dev_*.h headers.
 */
#pragma once
#include "dev_path.h"
#include "dev_trace.h"
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdio>
#include <mutex>

#define TPB 256
static_assert(TPB == LOBE_STRIDE, "the lobe columns in LDS are strided by the block size (dev_bsdf.h)");
#ifndef PT_WAVES
#define PT_WAVES 5 /* wavefronts per SIMD the persistent tracer is compiled for */
#endif
#define PT_WAVES_WIDE 4 /* ... and its second build, with 128 registers, for scenes with many analytic primitives (DScene::trace_waves) */
#define PT_WAVES_LEAN 6 /* ... and its third, without the SDF shapes' code, for scenes that have none: 80 registers (15-21 spilled), six blocks per CU */

struct Pipe {
    float *ro, *rd;      /* 3 planes of cap floats each */
    int4 *hit;  /* per path slot: (primitive or -1, face / item, t as bits, -) - one 16-byte store per ray of the tracer (round 3; three
                 * scattered 4-byte stores before: a persistent lane finishes its ray alone, and each was a partial write of its own) */
    int *route; /* per place in the bounce's queue of shaded vertices: where the vertex goes next (ROUTE_*, SHADE_DEFERRED), k_shade -> k_route */
    uint32_t *paths; /* PathRec as PATH_WORDS tiled planes (dev_nee.h: plane_at) */
    uint32_t *nee;   /* NeeRec as NEE_WORDS tiled planes */
    int cap;
};
struct PassParams {
    uint64_t seed;
    int npix, width, height;       /* npix: pixels THIS pass renders (all of the film, or the tiles of this context) */
    int sample_first, sample_step; /* sample of slot = sample_first + (slot / npix) * sample_step */
    int integrator;
    const int *pix_map;            /* pixel of the film for each pixel of the pass, or null: the same index (ptr_film_tiles) */
    int nee_p1;                    /* the pass resolves direct light with the in-kernel walks: k_shade stores the light point (NeeRec::p1*) too */
};
/* the film pixel a path slot belongs to: keys its RNG stream and places its camera ray */
__device__ __forceinline__ int pass_pixel(const PassParams &prm, int slot) {
    const int lp = slot % prm.npix;
    return prm.pix_map ? prm.pix_map[lp] : lp;
}


/* Path and direct-light records live in HBM as planes of 32-bit words (structure of arrays):
 * consecutive lanes work on consecutive slots, so every load / store of a field is one coalesced
 * request per wavefront; in registers a record is the plain struct. */
#define PATH_WORDS ((int)(sizeof(PathRec) / 4))
#define NEE_WORDS ((int)(sizeof(NeeRec) / 4))
#define WORD_OF(T, field) ((int)(offsetof(T, field) / 4))
template <int W0, int W1, typename T> PD void soa_load(const uint32_t *base, int cap, int slot, T *rec) {
    uint32_t *o = reinterpret_cast<uint32_t *>(rec);
#pragma unroll
    for (int w = W0; w < W1; w++) o[w] = base[plane_at((int)(sizeof(T) / 4), slot, w)];
}
template <int W0, int W1, typename T> PD void soa_store(uint32_t *base, int cap, int slot, const T *rec) {
    const uint32_t *o = reinterpret_cast<const uint32_t *>(rec);
#pragma unroll
#if PT_NT_RECORDS >= 2
    for (int w = W0; w < W1; w++) __builtin_nontemporal_store(o[w], &base[plane_at((int)(sizeof(T) / 4), slot, w)]);
#else
    for (int w = W0; w < W1; w++) base[plane_at((int)(sizeof(T) / 4), slot, w)] = o[w];
#endif
}
#define NEE_A_BEGIN WORD_OF(NeeRec, fA)
#define NEE_A_END WORD_OF(NeeRec, fB)
#define NEE_B_BEGIN WORD_OF(NeeRec, fB)
#define NEE_B_END WORD_OF(NeeRec, occA)
#define NEE_RES_BEGIN WORD_OF(NeeRec, occA)
#define NEE_RES_END (WORD_OF(NeeRec, tA) + 1)
#define NEE_EXTRA_BEGIN WORD_OF(NeeRec, extra)
#define NEE_EXTRA_END (WORD_OF(NeeRec, extra) + 3)
PD void nee_load(const Pipe &pp, int slot, NeeRec *rec) { /* header, then only the halves in use */
    soa_load<0, 4>(pp.nee, pp.cap, slot, rec);
    if (rec->flags & NEE_A) soa_load<NEE_A_BEGIN, NEE_A_END>(pp.nee, pp.cap, slot, rec);
    if (rec->flags & NEE_B) soa_load<NEE_B_BEGIN, NEE_B_END>(pp.nee, pp.cap, slot, rec);
    if (rec->flags & NEE_EXTRA) soa_load<NEE_EXTRA_BEGIN, NEE_EXTRA_END>(pp.nee, pp.cap, slot, rec);
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
/* The counters exist CTR_REPLICAS times, each copy in a 128-byte line of its own, and a block adds to the copy
 * its index selects: thousands of wavefronts flushing into ONE line at the end of every launch queue up behind
 * each other (atomics on a line are served one at a time).  ptr_counters_read sums the copies. */
#define CTR_REPLICAS 32
#define CTR_STRIDE 32 /* unsigned long longs per copy: two 128-byte lines, >= CX_END counters */
__device__ __forceinline__ void ctr_flush(unsigned long long *ctrs, int idx, unsigned v) {
    unsigned s = wave_sum(v);
    if (__lane_id() == 0 && s) atomicAdd(&ctrs[(blockIdx.x & (CTR_REPLICAS - 1)) * CTR_STRIDE + idx], (unsigned long long)s);
}
/* wave-aggregated append: one atomic per wavefront */
__device__ __forceinline__ void queue_append(int *counter, int *queue, bool pred, int value) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0) return;
    int lane = __lane_id();
    int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader, 64);
    if (pred) {
        unsigned long long lower = mask & ((1ull << lane) - 1ull);
        queue[base + __popcll(lower)] = value;
    }
}


enum { C_TOTAL = 0, C_MISS, C_SURF, C_MED, C_LIGHT, C_ZERO, C_SHADOW, C_MIS, C_PROBE, C_CAMERA, C_NODES, C_PRIMS, C_COUNT };
/* pipeline statistics kept in the same lines, behind the PtrCounters fields (ptr_pipeline_stats_read) */
enum { CX_FINISH_PATHS = C_COUNT, CX_FINISH_NEE, CX_SHADED, CX_NEE_RECORDS, CX_ANY_NODES, CX_ANY_ITEMS, CX_ANY_RAYS, CX_FAST_DEFERRED, CX_END };
static_assert(CX_END <= CTR_STRIDE, "a counter copy holds every counter");

/* grid-stride helper: every lane of a wavefront runs the same number of rounds, so the ballots of
 * queue_append / ctr_flush stay convergent; n may live in device memory (written by an earlier
 * kernel of the same stream) */
#define GRID_STRIDE_BEGIN(n_ptr, n_host)                                                           \
    const int n = (n_ptr) ? *(n_ptr) : (n_host);                                                   \
    for (int base_ = blockIdx.x * TPB + (threadIdx.x & ~63); base_ < n; base_ += gridDim.x * TPB) { \
        const int i = base_ + (threadIdx.x & 63);
#define GRID_STRIDE_END }

#ifndef SHADE_WAVES
#define SHADE_WAVES 2
#endif

/* More than 64 KB of dynamic LDS has to be asked for with hipFuncSetAttribute - per kernel AND per device (a RenderGroup runs
 * several GPUs from one process, its drivers may launch from one host thread per context): `st` is the launcher's own
 * static state, one bit per device, taken under its mutex; a failure is reported and retried at the next launch. */
struct DynLdsOnce {
    std::mutex mu;
    unsigned long long done = 0;
};
static inline void ensure_max_dynamic_lds(DynLdsOnce &st, const void *kernel, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
    std::lock_guard<std::mutex> lk(st.mu);
    if ((st.done >> dev) & 1ull) return;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) st.done |= 1ull << dev;
    else fprintf(stderr, "pathtracer_amd: hipFuncSetAttribute(MaxDynamicSharedMemorySize, %d) on device %d: %s\n", bytes, dev, hipGetErrorString(e));
}

/* the probe-chain records of a lane (dev_bssrdf.h, rt_shade.h): planes == null when the pass runs without them */
struct SssChains {
    uint32_t *planes; /* SSS_WORDS tiled planes, `cap` records */
    int cap;
    int *n_chains;    /* device counter of the bounce: deferred vertices numbered so far (may exceed cap) */
    int *n_live0;     /* ... and how many of them got a record: chains 0 .. n_live0-1 enter the first round */
    ExitNee exit;     /* direct light at the exit points as records of the tracer's launches (dev_bssrdf.h), or {null, null} */
};

/* ---- launchers (one per kernel translation unit) ---- */
struct ShadeArgs {
    DScene sc;
    Pipe pp;
    PassParams prm;
    const int *n_ptr;
    int n_host;
    const int *queue;
    SssChains chains;
    int walk;
    int max_lobes; /* SSS kernel only: lobes of LDS per vertex (the scene's maximum), see rt_shade.h */
    unsigned long long *ctrs;
};
struct FinishArgs {
    DScene sc;
    Pipe pp;
    PassParams prm;
    const int *n_ptr;
    int n_host;
    const int *queue;
    int walk;
    int max_lobes; /* lobes of LDS per path: the scene's maximum (rt_finish.h) */
    unsigned long long *ctrs;
};
void rt_launch_shade(unsigned blocks, hipStream_t s, const ShadeArgs &a);      /* rt_shade.hip */
void rt_launch_shade_sss(unsigned blocks, hipStream_t s, const ShadeArgs &a);  /* rt_shade_sss.hip */
void rt_launch_shade_defer(unsigned blocks, hipStream_t s, const ShadeArgs &a); /* rt_shade_defer.hip */
void rt_launch_shade_begin(unsigned blocks, hipStream_t s, const ShadeArgs &a); /* rt_shade_defer.hip */
void rt_launch_shade_tail(unsigned blocks, hipStream_t s, const ShadeArgs &a);  /* rt_shade_tail.hip */
/* every probe chain of a bounce traced to its end in one launch of the continuing persistent tracer (rt_trace.hip) */
void rt_launch_pchain(unsigned blocks, hipStream_t s, const DScene &sc, uint32_t *planes, const int *n_ptr, int *cursor, unsigned long long *ctrs);
void rt_launch_finish(unsigned blocks, hipStream_t s, const FinishArgs &a);     /* rt_finish.hip */
void rt_launch_finish_sss(unsigned blocks, hipStream_t s, const FinishArgs &a); /* rt_finish_sss.hip */
void rt_launch_shadow_walk(unsigned blocks, hipStream_t s, const DScene &sc, const Pipe &pp, const int *n_ptr, int n_host, const int *queue,
                           unsigned long long *ctrs); /* rt_walk.hip */
void rt_launch_mis_walk(unsigned blocks, hipStream_t s, const DScene &sc, const Pipe &pp, const int *n_ptr, int n_host, const int *queue,
                        const int *q_hit, unsigned long long *ctrs);
void rt_launch_ptrace2any(unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &a, const TraceIO &b);
void rt_launch_ptrace3(bool count, unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &path, const TraceIO &shadow, const TraceIO &mis,
                       unsigned long long *ctrs);
void rt_launch_ptrace(bool any, bool count, unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &io, unsigned long long *ctrs); /* rt_trace.hip */
bool rt_ptrace_is_fast(bool any, const DScene &sc, const TraceIO &io); /* would that launch walk the fast top-level tree? */
/* the same three launches with four lanes per ray (dev_qtrace.h), rt_qtrace.hip */
void rt_launch_qtrace2any(unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &a, const TraceIO &b);
void rt_launch_qtrace3(bool count, unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &path, const TraceIO &shadow, const TraceIO &mis,
                       unsigned long long *ctrs);
void rt_launch_qtrace(bool any, bool count, unsigned blocks, hipStream_t s, const DScene &sc, const TraceIO &io, unsigned long long *ctrs);
/* unit hooks (rt_unit.hip) */
void rt_launch_unit(hipStream_t s, const DScene &sc, int prim, int kind, int n, const float *in, int in_w, float *out, int out_w);
void rt_launch_unit_bsdf(hipStream_t s, const DScene &sc, int prim, int sample, int n, const float *wo, const float *wi_or_u, const float *uv, float *wi_out,
                         float *f, float *pdf, int *flags);
