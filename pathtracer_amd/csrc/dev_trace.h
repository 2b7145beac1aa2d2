/*
 * dev_trace.h — persistent-wavefront ray tracer: the scene query of Aggregator::Intersect
 * (primitive.cpp:374-431) for a queue of rays, with the per-ray logic of traverse<>() in
 * dev_isect.h unrolled into a per-lane state machine.
 *
 * Why: ray cost on this path is heavy-tailed (config[1]: median 4 box tests, mean 11-27, 99.9th
 * percentile ~700, maximum ~1500; a wavefront that keeps its 64 rays until the last one ends
 * spends ~20x the mean).  Here a wavefront is persistent: a lane whose ray has ended takes the
 * next ray from the queue (64 indices per atomic, handed out by ballot rank), so a long ray
 * occupies one lane, not a wavefront.  Each ray still sees exactly the operations of traverse<>:
 * top level = the reference's own primitive BVH walked in its order, geometry level = 4-wide SAH
 * BVH, nearest child first, closest-hit rule t <= tMax.
 */
#pragma once
#include "dev_isect.h"
#include "dev_nee.h"

struct TraceIO {
    const int *queue; /* ray slots, or null for identity */
    int n;
    const int *n_ptr; /* if set, the ray count is read from device memory (written earlier in the stream) */
    /* ray origin / direction: component c of slot s at base + s*es + c*cs (bytes): SoA planes or AoS */
    const char *o, *d;
    long long o_es, o_cs, d_es, d_cs;
    const char *tmax; /* per-slot float, or null: tmax_const */
    long long tmax_es;
    float tmax_const;
    /* outputs, any may be null.  Closest: prim (-1 = miss), face, t (tmax passes through on a
     * miss), b0, b1.  ANY: prim receives 1 / 0 (occluded or not). */
    char *prim, *face, *t, *b0, *b1;
    long long prim_es, face_es, t_es, b0_es, b1_es;
    int *counter; /* ray-fetch cursor, zero at launch */
    int n_cursors; /* > 1: that many cursors, PT_CURSOR_STRIDE ints apart from `counter` on (all zero at launch), each over its own stretch
                    * of the queue - atomics on one address are served one at a time, on different cache lines side by side (persistent_trace) */
    /* rays and results inside tiled word planes (dev_nee.h plane_at; the direct-light records):
     * tile_words = words per record (0: the byte strides above apply), o / prim = the plane base,
     * w_o, w_d, w_prim, w_t = first word of origin, direction, prim result, t result (-1: none) */
    int tile_words, w_o, w_d, w_prim, w_t;
    int any_as_hit; /* any-hit launch whose result is read like a closest hit's prim word: some prim id / -1 */
    int w_face1;    /* tiled planes, closest hit: 1 + the word that receives the face (0: the face is not stored) */
    /* fast top level (persistent_trace<.., FAST = true>): the closest-hit rays whose answer may need the reference's order are
     * left here - their slots, counted in *fb_count - for a second launch that walks the reference's tree */
    int4 *hit16; /* closest hit: if set, the whole result of slot s goes to hit16[s] = (prim, face, t, 0) in one store (the pipeline's path segments) */
    int *fb_queue, *fb_count;
    int is_fallback; /* this IS that second launch (its ray count goes to the CX_FAST_DEFERRED statistic) */
};

/* persistent_trace<..., PAIRS>: how the leaf phase runs - an item loop per lane (0), or (ray, triangle) pairs dealt to all lanes:
 * PT_PAIRS_TRI for scenes without particle clouds, PT_PAIRS_MIXED with the item loop kept for the leaves of clouds */
enum { PT_PAIRS_OFF = 0, PT_PAIRS_TRI = 1, PT_PAIRS_MIXED = 2 };
#define GEOM_BOTTOM INT32_MIN /* stack entry under a mesh's / cloud's own (= BVH_NO_CHILD, which is never pushed) */
enum { TM_IDLE = 0, TM_TOP = 1, TM_TOPLEAF = 2, TM_GEOM = 3, TM_DONE = 4 /* CONTINUES sources: the ray has its answer and waits for src.advance */ };

#define PT_PUSH(v)                                          \
    do {                                                    \
        if (sp < LDS_STACK) lds[sp * TRACE_TPB] = (v);      \
        else if (sp < BVH_STACK) ovf[sp - LDS_STACK] = (v); \
        sp++;                                               \
    } while (0)
#define PT_POP() (--sp, sp < LDS_STACK ? lds[sp * TRACE_TPB] : ovf[(sp < BVH_STACK ? sp : BVH_STACK - 1) - LDS_STACK])

/* inner loops hand control back (to let finished lanes refill) once this few lanes remain in them */
#ifndef PT_MIN_ACTIVE
#define PT_MIN_ACTIVE 16 /* (8 before the leaf phase ran in pair mode: a leaf round is now cheap, leaving the node loop early costs less) */
#endif
/* ... in the build for scenes of many analytic primitives (the one with the wider top-level batch, rt_trace.hip), where few lanes are
 * inside a mesh at any time: configs[2] 610 ms per step with 8, 625 with 16 */
#ifndef PT_MIN_ACTIVE_WIDE
#define PT_MIN_ACTIVE_WIDE 8
#endif
/* rays a wavefront reserves per atomic on the queue cursor */
#ifndef PT_POOL
#define PT_POOL 128 /* every wavefront of every launch adds to ONE cursor, and atomics on one address are served one at a time (~10 ns): at 64 rays per atomic the 361 K atomics of a 23 M-ray launch were 3.6 of its 5.5 ms.  configs[1] closest-hit launches per pass: 32 -> 55 ms, 64 -> 32.9, 128 -> 24.5, 256 -> 24.8, 512 -> 26.6, 1024 -> 30.5 (the rays a wavefront holds back at the end of a launch) */
#endif
/* lanes that want a top-level step wait until this many do; the build for scenes with many analytic primitives (rt_trace.hip,
 * PT_WAVES_WIDE) waits for more: there the top level is most of a ray's work and is worth running full (configs[2]: 649 -> 620 ms
 * per step with 32 instead of 16; configs[1], where it is a sphere and a mesh: 133 -> 139) */
#ifndef PT_TOP_BATCH
#define PT_TOP_BATCH 16
#endif
#ifndef PT_TOP_BATCH_WIDE
#define PT_TOP_BATCH_WIDE 32
#endif
/* pair mode: the leaf phase runs another round at once while this many pairs wait (0: one round per outer iteration) */
#ifndef PT_LEAF_REPEAT
#define PT_LEAF_REPEAT 16 /* configs[1], one pass alone: 125.3 ms with 0, 120.7 with 16 or 32, 121.0 with 48, 121.8 with 1 */
#endif
#define PT_CURSOR_STRIDE 16 /* ints between the cursors of a launch: a cache line each */
/* the part of a launch's queue that is dealt out to the wavefronts without atomics (persistent_trace) */
#ifndef PT_STATIC_NUM
#define PT_STATIC_NUM 0 /* measured, configs[1], closest-hit launches per pass: 0 -> 25.5 ms, 1/2 -> 25.4, 3/4 -> 26.6, 7/8 -> 27.8 (pool 128); 3/4 -> 27.3, 15/16 -> 28.7 (pool 64): what the wavefronts' shares cost differs by more than the rest evens out */
#endif
#ifndef PT_STATIC_DEN
#define PT_STATIC_DEN 4
#endif
/* top level: a leaf that holds nothing more is left, and a walk whose stack has run out is ended, in the step that finds it so
 * instead of in steps of their own (6 -> 3-4 wave iterations per ray in a scene of two primitives; same tests, same order, films
 * identical).  0 never, 1 always, 2 in the build for scenes of many analytic primitives only (the one with the wider top-level batch):
 * configs[2] 593.8 -> 585.5 ms per step, configs[0] and [5] equal, but configs[1] 104.0 -> 109.2 ms per pass (the any-hit launches
 * +2 ms each way: lanes that leave a mesh join the top-level batch a step earlier and it runs with fewer of them) */
#ifndef PT_TOP_FUSE
#define PT_TOP_FUSE 2
#endif
/* idle lanes wait until this many can be refilled together */
#ifndef PT_REFILL
#define PT_REFILL 24
#endif

PD void trace_io_fetch(const TraceIO &io, int idx, int *slot, f3 *ro, f3 *rd, float *tmax) {
    const int s = io.queue ? io.queue[idx] : idx;
    if (io.tile_words) {
        const float *b = reinterpret_cast<const float *>(io.o);
        const size_t step = (size_t)1 << PLANE_TILE_SHIFT, io0 = plane_at(io.tile_words, s, io.w_o), id0 = plane_at(io.tile_words, s, io.w_d);
        *ro = f3(b[io0], b[io0 + step], b[io0 + 2 * step]);
        *rd = f3(b[id0], b[id0 + step], b[id0 + 2 * step]);
        *tmax = io.tmax_const;
        *slot = s;
        return;
    }
    const char *po = io.o + (long long)s * io.o_es;
    const char *pd = io.d + (long long)s * io.d_es;
    *ro = f3(*(const float *)po, *(const float *)(po + io.o_cs), *(const float *)(po + 2 * io.o_cs));
    *rd = f3(*(const float *)pd, *(const float *)(pd + io.d_cs), *(const float *)(pd + 2 * io.d_cs));
    *tmax = io.tmax ? *(const float *)(io.tmax + (long long)s * io.tmax_es) : io.tmax_const;
    *slot = s;
}
template <bool COUNT>
PD void trace_io_store(const TraceIO &io, bool any, int slot, bool hit, int hp, int hf, float tmax, float hb0, float hb1, unsigned rn, unsigned ri) {
    if (io.hit16) {
        io.hit16[slot] = make_int4(hit ? hp : -1, hit ? hf : -1, __float_as_int(tmax), 0);
        return;
    }
    if (io.tile_words) {
        int *b = reinterpret_cast<int *>(io.prim);
        b[plane_at(io.tile_words, slot, io.w_prim)] = (any && !io.any_as_hit) ? (hit ? 1 : 0) : (hit ? hp : -1);
        if (!any && io.w_t >= 0) b[plane_at(io.tile_words, slot, io.w_t)] = __float_as_int(tmax);
        if (!any && io.w_face1 > 0) b[plane_at(io.tile_words, slot, io.w_face1 - 1)] = hit ? hf : -1;
        return;
    }
    if (any) {
        if (io.prim) *(int *)(io.prim + (long long)slot * io.prim_es) = hit ? 1 : 0;
        return;
    }
    if (io.prim) *(int *)(io.prim + (long long)slot * io.prim_es) = hit ? hp : -1;
    if (io.face) *(int *)(io.face + (long long)slot * io.face_es) = hit ? hf : -1;
    if (io.t) *(float *)(io.t + (long long)slot * io.t_es) = tmax;
    if (COUNT && io.b0 && io.b1) {
        *(float *)(io.b0 + (long long)slot * io.b0_es) = (float)rn;
        *(float *)(io.b1 + (long long)slot * io.b1_es) = (float)ri;
    } else {
        if (io.b0) *(float *)(io.b0 + (long long)slot * io.b0_es) = hit ? hb0 : 0;
        if (io.b1) *(float *)(io.b1 + (long long)slot * io.b1_es) = hit ? hb1 : 0;
    }
}

/* Where the rays of a launch come from.  OneSource: one queue, all closest-hit or all any-hit
 * (known at compile time).  ThreeSources: the three queues a bounce produces - next path segments
 * (closest hit), shadow rays (any hit), MIS rays (closest hit) - served by ONE launch: a launch
 * lasts about as long as its longest ray, so three launches per bounce pay that tail three times
 * and run their last waves on a mostly empty GPU. */
template <bool ANY, bool WITH_BARY = false>
struct OneSource {
    static constexpr bool CONTINUES = false;
    static constexpr bool BARY = WITH_BARY; /* the launch stores barycentrics (the stand-alone ptr_trace_closest; the pipeline rebuilds its hits from prim / face / t) */
    TraceIO io;
    int n_rays;
    int *counter;
    int n_cursors;
    PD void init() { n_rays = io.n_ptr ? *io.n_ptr : io.n; counter = io.counter; n_cursors = io.n_cursors; }
    PD int total() const { return n_rays; }
    PD bool is_any(int) const { return ANY; }
    PD void fetch(int idx, int *tag, int *slot, f3 *ro, f3 *rd, float *tmax) const { *tag = 0; trace_io_fetch(io, idx, slot, ro, rd, tmax); }
    template <bool COUNT>
    PD void store(int, int slot, bool hit, int hp, int hf, float tmax, float hb0, float hb1, unsigned rn, unsigned ri) const {
        trace_io_store<COUNT>(io, ANY, slot, hit, hp, hf, tmax, hb0, hb1, rn, ri);
    }
    PD void defer(int, int slot) const { io.fb_queue[atomicAdd(io.fb_count, 1)] = slot; } /* (rare: one atomic per ray is fine) */
};
struct ThreeSources {
    static constexpr bool CONTINUES = false;
    static constexpr bool BARY = false;
    TraceIO path, shadow, mis; /* path.counter is the fetch cursor of the launch */
    int n0, n01, n012;
    int *counter;
    int n_cursors;
    PD void init() {
        n0 = path.n_ptr ? *path.n_ptr : path.n;
        n01 = n0 + (shadow.n_ptr ? *shadow.n_ptr : shadow.n);
        n012 = n01 + (mis.n_ptr ? *mis.n_ptr : mis.n);
        counter = path.counter;
        n_cursors = path.n_cursors;
    }
    PD int total() const { return n012; }
    PD bool is_any(int tag) const { return tag == 1; }
    PD void fetch(int idx, int *tag, int *slot, f3 *ro, f3 *rd, float *tmax) const {
        if (idx < n0) { *tag = 0; trace_io_fetch(path, idx, slot, ro, rd, tmax); }
        else if (idx < n01) { *tag = 1; trace_io_fetch(shadow, idx - n0, slot, ro, rd, tmax); }
        else { *tag = 2; trace_io_fetch(mis, idx - n01, slot, ro, rd, tmax); }
    }
    template <bool COUNT>
    PD void store(int tag, int slot, bool hit, int hp, int hf, float tmax, float hb0, float hb1, unsigned rn, unsigned ri) const {
        if (tag == 0) trace_io_store<COUNT>(path, false, slot, hit, hp, hf, tmax, hb0, hb1, rn, ri);
        else if (tag == 1) trace_io_store<COUNT>(shadow, true, slot, hit, hp, hf, tmax, hb0, hb1, rn, ri);
        else trace_io_store<COUNT>(mis, false, slot, hit, hp, hf, tmax, hb0, hb1, rn, ri);
    }
    PD void defer(int, int) const {} /* (never launched with FAST) */
};

/* Two any-hit queues in one launch: the shadow rays and - where any hit answers them (ptr_runtime.hip) - the MIS rays of a bounce */
struct TwoAny {
    static constexpr bool CONTINUES = false;
    static constexpr bool BARY = false;
    TraceIO a, b; /* a.counter is the fetch cursor of the launch */
    int n0, n01;
    int *counter;
    int n_cursors;
    PD void init() {
        n0 = a.n_ptr ? *a.n_ptr : a.n;
        n01 = n0 + (b.n_ptr ? *b.n_ptr : b.n);
        counter = a.counter;
        n_cursors = a.n_cursors;
    }
    PD int total() const { return n01; }
    PD bool is_any(int) const { return true; }
    PD void fetch(int idx, int *tag, int *slot, f3 *ro, f3 *rd, float *tmax) const {
        if (idx < n0) { *tag = 0; trace_io_fetch(a, idx, slot, ro, rd, tmax); }
        else { *tag = 1; trace_io_fetch(b, idx - n0, slot, ro, rd, tmax); }
    }
    template <bool COUNT>
    PD void store(int tag, int slot, bool hit, int hp, int hf, float tmax, float hb0, float hb1, unsigned rn, unsigned ri) const {
        if (tag == 0) trace_io_store<COUNT>(a, true, slot, hit, hp, hf, tmax, hb0, hb1, rn, ri);
        else trace_io_store<COUNT>(b, true, slot, hit, hp, hf, tmax, hb0, hb1, rn, ri);
    }
    PD void defer(int, int) const {} /* (any-hit rays are never deferred) */
};

/* A source whose rays CONTINUE: when a ray has its answer the lane does not take the next ray of the queue but waits (TM_DONE)
 * until a batch of lanes can run src.advance together - the per-ray work between two dependent scene queries - and then
 * traces the follow-up ray advance hands back, until it says the chain has ended.  A chain of dependent queries thus runs
 * to its end at the tracer's occupancy, in one launch, however long it is.
 * ChainSource: the BSSRDF probe chains (dev_bssrdf.h): ray = the chain's current segment, advance = chain_advance. */
struct ChainSource {
    static constexpr bool CONTINUES = true;
    static constexpr bool BARY = false;
    uint32_t *planes;
    const int *n_ptr;
    int n_rays;
    int *counter;
    static constexpr int n_cursors = 1;
    unsigned *probes; /* the kernel's per-lane count of probe rays */
    PD void init() { n_rays = *n_ptr; }
    PD int total() const { return n_rays; }
    PD bool is_any(int) const { return false; }
    PD void fetch(int idx, int *tag, int *slot, f3 *ro, f3 *rd, float *tmax) const {
        const ChainRef c = {planes, idx};
        *tag = 0;
        *slot = idx;
        *ro = c.get3(SW_O); /* a chain that was finished when it was made has a null segment: the ray ends at once, advance declines */
        *rd = c.get3(SW_D);
        *tmax = 1 - kShadowEpsilon; /* Interaction::SpawnRayTo(Point3f), interaction.cpp:12-16 */
    }
    template <bool COUNT>
    PD void store(int, int, bool, int, int, float, float, float, unsigned, unsigned) const {}
    PD bool advance(const DScene &sc, int, int slot, bool hit, int hp, int hf, float t, f3 *ro, f3 *rd, float *tmax) {
        const ChainRef c = {planes, slot};
        if (c.geti(SW_STATE) != SSS_LIVE) return false;
        (*probes)++; /* the segment just traced: one probe ray (bssrdf.cpp:536) */
        *tmax = 1 - kShadowEpsilon;
        return chain_advance(sc, c, hit ? hp : -1, hf, t, ro, rd);
    }
    PD void defer(int, int) const {} /* (never launched with FAST: a probe segment is not a unit vector) */
};

/* PT_STATS (diagnostic build only, tools/gpu_tracestats.py): where the lanes and the cycles of the persistent tracer go.
 * Per phase: wave iterations, lanes active in them, shader cycles between the phase's first and last instruction. */
#ifdef PT_STATS
enum { PS_OUTER = 0, PS_REFILL_IT, PS_REFILL_LANES, PS_TOP_IT, PS_TOP_LANES, PS_IN_IT, PS_IN_LANES, PS_LEAF_IT, PS_LEAF_LANES, PS_LEAF_ENTRY, PS_LEAF_ENTRY_LANES,
       PS_CYC_REFILL, PS_CYC_TOP, PS_CYC_IN, PS_CYC_LEAF, PS_CYC_TOTAL, PS_RAYS, PS_CYC_TOPPRIM, PS_TOPPRIM_IT, PS_TOPPRIM_LANES, PS_N };
static __device__ unsigned long long g_pt_stats[PS_N];
#define PS_ADD(k, v) do { st_[k] += (v); } while (0)
/* inside divergent code: one active lane adds for the wave */
#define PS_WAVE_IT(kit, klanes)                                                         \
    do {                                                                                \
        const unsigned long long m_ = __ballot(true);                                   \
        if (lane == __ffsll((long long)m_) - 1) { atomicAdd(&g_pt_stats[kit], 1ull); atomicAdd(&g_pt_stats[klanes], (unsigned long long)__popcll(m_)); } \
    } while (0)
#define PS_CLOCK() clock64()
#else
#define PS_ADD(k, v) do { } while (0)
#define PS_WAVE_IT(kit, klanes) do { } while (0)
#define PS_CLOCK() 0ll
#endif

/* a node visit of the geometry level: from the quantised nodes (dev_scene.h NodeQ) unless the build stages upper levels in LDS */
#if PT_QNODES && PT_LDS_TOP == 0
#define PT_VISIT(node, r0, r1, r2, r3) bvh4_visit_q(sc.qnodes, node, ro, inv, tmax, r0, r1, r2, r3, COUNT ? &rn : nullptr)
#else
#define PT_VISIT(node, r0, r1, r2, r3) bvh4_visit_staged(sc.nodes, staged, node, ro, inv, tmax, r0, r1, r2, r3, COUNT ? &rn : nullptr)
#endif
/* Postponed leaves (Aila & Laine's speculative traversal): a lane that reaches a leaf keeps it aside and walks on - with the
 * tMax it has - until it reaches the next one, so that more lanes are in the inner-node loop while it runs and more of them
 * bring a leaf when it ends.  Leaves are still looked into in the order they were met; a node visited before the postponed
 * leaf has shrunk tMax is visited with the older, larger tMax: more boxes entered, no candidate lost. */
#ifndef PT_SPEC
#define PT_SPEC 0 /* measured (round 3): films identical, closest-hit launch 6.53-6.55 ms against 6.45-6.46 without - off */
#endif
#ifndef PT_PREFETCH
#define PT_PREFETCH 0
#endif
/* ---- fast top level (round 3; FAST = true) ----------------------------------------------------------------------
 * A scene of more than a handful of primitives carries a second tree over them, built for speed (ptr_build.cpp
 * build_fast_top; DScene::top_fast): in a room of a hundred primitives a ray steps through ~25 nodes of the reference's
 * median-split tree with its random axes.  A ray walks the fast tree when its answer provably does not need the reference's order:
 *   any hit      always: tMax does not change before the walk ends, so every primitive is tested with the same tMax and
 *                reached through the same test of its own bounds (made here at the leaf, as the reference's one-primitive
 *                leaves make it) whatever the order;
 *   closest hit  if its direction is a unit vector to rounding (an SDF shape measures t along the normalised direction,
 *                procedural.cpp:238: for other rays its answer can EXCEED the tMax it was given, and then everything
 *                depends on order).  The walk keeps tbest = the smallest accepted t and tests everything against the
 *                wider bound tmax = G = max(entry distance of the winner's own bounds, tbest * (1 + 1e-5)).  The
 *                reference's walk accepts a hit when t <= tMax and its bounds are entered before tMax, and its tMax only
 *                falls: it returns the candidate of smallest t - unless it culled that candidate by its bounds (possible
 *                only if another candidate lies in front of those bounds: t < entry distance), or two candidates are equal
 *                in t (the later wins).  Both need a second candidate with t < G, and every way such a candidate can show
 *                up defers the ray: an accepted t above tbest, a superseded tbest below the new G, a hit whose own bounds
 *                lie behind the bound of the walk, a mesh winner whose bounds lie behind its own hit.  (What was rejected
 *                or culled under an earlier bound lies beyond that bound, which was at least the tbest of the time - so if
 *                it lies below the final G, so does a superseded tbest.)
 * A deferred ray's slot goes to the launch's fallback queue (TraceIO::fb_queue); a second launch (FAST = false) walks those rays
 * through the reference's tree from their start: the same answer as ever, for every ray.  The fast tree's boxes are wider than
 * the primitives' bounds (they hold every hit point a primitive can report), so culling by them loses no candidate. */
#define FZ_WIDEN 1.00001f
template <class Source, bool COUNT, bool FAST = false, int TOP_STAGED = TOP_LDS_NODES, int TOP_BATCH = PT_TOP_BATCH, bool PROC = true, int PAIRS = 0>
PD void persistent_trace(const DScene &sc, Source src, int *lds, const float4 *staged, const float4 *top_staged, unsigned *total_nodes,
                         unsigned *total_items, unsigned char *pairs = nullptr) {
    int ovf[OVF_STACK];
#ifdef PT_STATS
    unsigned long long st_[PS_N] = {};
    const long long t_begin_ = clock64();
#endif
    const int lane = __lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    constexpr bool TOP_FUSE = PT_TOP_FUSE == 1 || (PT_TOP_FUSE == 2 && TOP_BATCH > PT_TOP_BATCH);

    int mode = TM_IDLE, slot = -1, tag = 0;
    src.init();
    f3 ro, rd, inv;
    TriRay tr;
    tr.kx = tr.ky = tr.kz = 0; tr.Sx = tr.Sy = tr.Sz = 0;
    float tmax = 0, hb0 = 0, hb1 = 0; /* (hb0, hb1: only sources with BARY keep them - two registers the pipeline's launches do without) */
    int hp = -1, hf = -1;
    bool hit = false;
    /* (in TM_TOPLEAF tnode holds the end of the leaf's handle range: a node is only named again by the pop that leaves the leaf) */
    int tnode = -1, leaf_pos = 0, cp = -1, node = 0, sp = 0;
#if PT_SPEC
    int pleaf = 0; /* a leaf the lane has met and not yet looked into (0: none): it walks on to its next leaf first, see the geometry level */
#endif
    unsigned rn = 0, ri = 0; /* per-ray work (COUNT) */
    bool in_cloud = false; /* PAIRS == PT_PAIRS_MIXED: the primitive the lane is inside is a particle cloud */
    bool fresh = false; /* top level: tnode was reached by descent from its parent, where its box has just passed with this tMax */
    float tbest = 0; /* FAST closest-hit walk: the smallest accepted t (tmax is then the slightly wider bound everything is tested with) */
    const int top_root = FAST ? sc.top_fast : 0;

    const int n_rays = src.total();
    int pool_pos = 0, pool_end = 0; /* wave-uniform */
    bool exhausted = false;
    /* The first PT_STATIC_NUM / PT_STATIC_DEN of the queue is dealt out without atomics: pool j * n_waves + w belongs to wavefront w
     * (interleaved, so that every wavefront sees every part of the image), the rest goes through the cursor and evens out what the
     * wavefronts' shares cost. */
    const int n_waves = (int)gridDim.x * (TRACE_TPB / 64), wave_id = (int)blockIdx.x * (TRACE_TPB / 64) + (int)(threadIdx.x >> 6);
    const int static_rounds = (int)(((long long)n_rays * PT_STATIC_NUM / PT_STATIC_DEN) / ((long long)n_waves * PT_POOL));
    const int static_total = static_rounds * n_waves * PT_POOL;
    int static_round = 0;
    /* several cursors (TraceIO::n_cursors): the rest of the queue in that many stretches of whole pools; a wavefront starts on stretch
     * wave_id mod n (so that the wavefronts of a block, and the blocks of a CU, spread over all of them) */
    const int n_cursors = src.n_cursors > 1 ? src.n_cursors : 1;
    const int seg_len = n_cursors > 1 ? (((n_rays - static_total) / n_cursors + PT_POOL) / PT_POOL) * PT_POOL : n_rays;
    int cur_seg = n_cursors > 1 ? wave_id % n_cursors : 0, segs_empty = 0, seg_end = n_rays; /* wave-uniform */
#if PT_PREFETCH
    int pf_acc = 0, pf_pending = 0; /* the first word of the nearest pushed sibling, loaded one visit ahead of its use */
#endif

#define PT_FINISH()                                                                  \
    do {                                                                             \
        if (COUNT) { *total_nodes += rn; *total_items += ri; }                       \
        if (Source::CONTINUES) mode = TM_DONE;                                       \
        else {                                                                       \
            src.template store<COUNT>(tag, slot, hit, hp, hf, tmax, hb0, hb1, rn, ri); \
            mode = TM_IDLE;                                                          \
            slot = -1;                                                               \
        }                                                                            \
    } while (0)
#define PT_ANY (src.is_any(tag))
#define PT_FZ_CLOSEST (FAST && !PT_ANY)
/* the ray goes to the fallback queue: its answer may depend on the reference's order */
#define PT_DEFER()                                                                   \
    do {                                                                             \
        if (COUNT) { *total_nodes += rn; *total_items += ri; }                       \
        src.defer(tag, slot);                                                        \
        mode = TM_IDLE;                                                              \
        slot = -1;                                                                   \
    } while (0)
/* the ray leaves a mesh / cloud: a hit it took there becomes the bound for what follows */
/* the walk has ended (the stack's bottom entry came up) */
#define PT_TOP_END()                                                                 \
    do {                                                                             \
        bool dfr_ = false;                                                           \
        if (PT_FZ_CLOSEST && hit) {                                                  \
            /* a mesh / cloud winner: would the reference have reached it?  (an analytic winner: G says so) */ \
            const DPrim &pw = sc.prims[hp];                                          \
            float te;                                                                \
            if ((pw.type == SH_MESH || pw.type == SH_CLOUD) &&                       \
                !slab_test6(pw.bmin[0], pw.bmin[1], pw.bmin[2], pw.bmax[0], pw.bmax[1], pw.bmax[2], ro, inv, tbest, &te)) dfr_ = true; \
            else tmax = tbest;                                                       \
        }                                                                            \
        if (dfr_) PT_DEFER();                                                        \
        else PT_FINISH();                                                            \
    } while (0)
#define PT_LEAVE_GEOM()                                                              \
    do {                                                                             \
        if (PT_FZ_CLOSEST && hit && hp == cp) tmax = tbest * FZ_WIDEN;               \
        if (TOP_FUSE && leaf_pos == tnode) { /* ... and its top-level leaf, which holds nothing more: the pop now, not in a step of its own */ \
            tnode = PT_POP();                                                        \
            fresh = false;                                                           \
            mode = TM_TOP;                                                           \
        } else mode = TM_TOPLEAF;                                                    \
    } while (0)

    for (;;) {
        PS_ADD(PS_OUTER, 1);
        const long long c0_ = PS_CLOCK();
        /* ---- CONTINUES sources: the rays that have their answer, in batches like the refill below ---- */
        if constexpr (Source::CONTINUES) {
            const unsigned long long done = __ballot(mode == TM_DONE);
            if (done && (__popcll(done) >= PT_REFILL || __ballot(mode == TM_GEOM) == 0)) {
                if (mode == TM_DONE) {
                    if (src.advance(sc, tag, slot, hit, hp, hf, tmax, &ro, &rd, &tmax)) {
                        inv = f3(1 / rd.x, 1 / rd.y, 1 / rd.z);
                        tr = tri_ray_setup(rd);
                        hp = -1; hf = -1; hb0 = 0; hb1 = 0; hit = false;
                        rn = 0; ri = 0;
                        sp = 0;
                        PT_PUSH(-1);
                        tnode = sc.n_top == 0 ? -1 : top_root; /* (an empty scene: the walk ends at its first step) */
                        fresh = false;
                        mode = TM_TOP;
                    } else {
                        mode = TM_IDLE;
                        slot = -1;
                    }
                }
            }
        }
        /* ---- refill idle lanes ---- */
        unsigned long long idle = __ballot(mode == TM_IDLE);
        /* refill in batches: the top-level phase a new ray starts with (analytic primitives: a
         * sphere costs a double-precision quadratic) should run with many lanes, not one or two */
        if (idle && (__popcll(idle) >= PT_REFILL || __ballot(mode == TM_GEOM) == 0)) {
            if (pool_pos >= pool_end && !exhausted) {
                int base = 0;
                if (static_round < static_rounds) { /* (wave-uniform) this wavefront's next pool of the dealt-out part: no atomic */
                    base = (static_round * n_waves + wave_id) * PT_POOL;
                    static_round++;
                } else if (n_cursors > 1) {
                    /* the stretch this wavefront is working on, then - once that one has been found empty - the next, until all have */
                    for (;;) {
                        int got = 0;
                        if (lane == 0) got = atomicAdd(src.counter + cur_seg * PT_CURSOR_STRIDE, PT_POOL);
                        got = __builtin_amdgcn_readfirstlane(got);
                        const int seg_begin = static_total + cur_seg * seg_len;
                        seg_end = seg_begin + seg_len < n_rays ? seg_begin + seg_len : n_rays;
                        base = seg_begin + got;
                        if (got < seg_len && base < seg_end) break;
                        base = n_rays;
                        if (++segs_empty >= n_cursors) break;
                        cur_seg = cur_seg + 1 == n_cursors ? 0 : cur_seg + 1;
                    }
                } else {
                    if (lane == 0) base = atomicAdd(src.counter, PT_POOL);
                    base = static_total + __shfl(base, 0, 64);
                }
                if (base >= n_rays) exhausted = true;
                else {
                    pool_pos = base;
                    pool_end = base + PT_POOL < seg_end ? base + PT_POOL : seg_end;
                }
            }
            const int avail = pool_end - pool_pos;
            if (avail > 0) {
                const int rank = __popcll(idle & lt_mask);
                if (mode == TM_IDLE && rank < avail) {
                    PS_WAVE_IT(PS_REFILL_IT, PS_REFILL_LANES);
                    const int idx = pool_pos + rank;
                    src.fetch(idx, &tag, &slot, &ro, &rd, &tmax);
                    inv = f3(1 / rd.x, 1 / rd.y, 1 / rd.z);
                    tr = tri_ray_setup(rd);
                    hp = -1; hf = -1; hb0 = 0; hb1 = 0; hit = false;
                    rn = 0; ri = 0;
                    sp = 0;
                    if (sc.n_top == 0) PT_FINISH();
                    else if (PT_FZ_CLOSEST && !(absr(dot(rd, rd) - 1.f) < 1e-5f)) PT_DEFER(); /* not a unit direction: the reference's tree */
                    else {
                        PT_PUSH(-1);
                        tnode = top_root;
                        fresh = false;
                        mode = TM_TOP;
                    }
                }
                const int want = __popcll(idle);
                pool_pos += want < avail ? want : avail;
            }
            if (exhausted && __ballot(mode != TM_IDLE) == 0) break; /* every wave reaches this: the grid drains */
        }

        /* ---- top level: the reference's own tree, step for step (primitive.cpp:374-431) ----
         * Lanes that come back from a mesh need a few more steps up here (leave the leaf, pop, maybe finish);
         * run for two or three lanes, those steps would hold up every lane that is inside a mesh.  So they wait
         * until PT_TOP_BATCH lanes want the top level (a refill brings PT_REFILL at once) or nobody is in a mesh. */
        const long long c1_ = PS_CLOCK();
        const unsigned long long in_top = __ballot(mode == TM_TOP || mode == TM_TOPLEAF);
        if (__popcll(in_top) >= TOP_BATCH || __ballot(mode == TM_GEOM) == 0)
        while (mode == TM_TOP || mode == TM_TOPLEAF) {
            PS_WAVE_IT(PS_TOP_IT, PS_TOP_LANES);
            int p = -1;            /* the primitive this iteration tests, if any */
            bool leaf_box = false; /* ... after its leaf's box, which is the primitive's own bounds */
            if (mode == TM_TOP) {
                if (tnode == -1) {
                    PT_TOP_END();
                    break;
                }
                if (tnode < -1) {
                    /* a leaf of one primitive, named by its parent (TopNode::pad): no record to read.  Reached by descent, its box
                     * has just passed at the parent with this same tMax; popped, it is tested again with the tMax of now, as the
                     * reference does (IntersectNode at the pop).  FAST: the primitive's bounds are tested below in any case. */
                    p = -tnode - 2;
                    leaf_box = !FAST && !fresh;
                    leaf_pos = tnode = 0;
                    mode = TM_TOPLEAF;
                } else {
                    /* one record per step: this node's box and, for an inner node, copies of both children's (TopNode);
                     * the upper levels of the tree this kernel's rays walk come from the block's LDS copy */
                    float4 q0, q1, q2, q3, q4, q5;
                    const int tl = tnode - top_root;
                    if (top_staged && tl < TOP_STAGED) {
                        const float4 *np = top_staged + tl * TOP_LDS_ROW;
                        q0 = np[0]; q1 = np[1]; q2 = np[2]; q3 = np[3]; q4 = np[4]; q5 = np[5];
                    } else {
                        const float4 *np = reinterpret_cast<const float4 *>(sc.top + tnode);
                        q0 = np[0]; q1 = np[1]; q2 = np[2]; q3 = np[3]; q4 = np[4]; q5 = np[5];
                    }
                    const int nd_left = __float_as_int(q0.w), nd_right = __float_as_int(q1.w);
                    float te;
                    /* (a node reached by descent passed this very test - same box, same ray, same tMax - one step ago, at its parent) */
                    bool enter = fresh;
                    if (!enter) {
                        if (COUNT) rn += 1;
                        enter = slab_test6(q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, ro, inv, tmax, &te);
                    }
                    fresh = false;
                    if (enter) {
                        if (nd_left < 0) {
                            leaf_pos = __float_as_int(q2.x);
                            tnode = leaf_pos + __float_as_int(q2.y);
                            mode = TM_TOPLEAF;
                        } else {
                            float tl0, tr0;
                            if (COUNT) rn += 2;
                            const bool visitL = slab_test6(q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, ro, inv, tmax, &tl0);
                            const bool visitR = slab_test6(q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, ro, inv, tmax, &tr0);
                            const int pl = __float_as_int(q2.z), pr_ = __float_as_int(q2.w);
                            const int ref_l = pl >= 0 ? -pl - 2 : nd_left, ref_r = pr_ >= 0 ? -pr_ - 2 : nd_right;
                            if (visitR && visitL) {
                                int first, second;
                                if (tr0 < tl0) { first = ref_r; second = ref_l; }
                                else { first = ref_l; second = ref_r; }
                                PT_PUSH(second);
                                tnode = first;
                                fresh = true;
                            } else if (visitL) { tnode = ref_l; fresh = true; }
                            else if (visitR) { tnode = ref_r; fresh = true; }
                            else tnode = PT_POP();
                        }
                    } else tnode = PT_POP();
                }
            }
            if (mode == TM_TOPLEAF) { /* (not `else`: a leaf step and its first primitive share an iteration) */
                if (p < 0) {
                    if (leaf_pos == tnode) {
                        tnode = PT_POP();
                        fresh = false;
                        mode = TM_TOP;
                        continue;
                    }
                    p = sc.order[leaf_pos++];
                }
                const DPrim &pr = sc.prims[p];
                const bool geom = pr.type == SH_MESH || pr.type == SH_CLOUD;
                if (leaf_box || (FAST && (geom || PT_ANY))) {
                    /* FAST: the fast tree's leaf boxes are wider than the primitive's bounds, which are what the reference's walk tests
                     * before it asks a primitive (its leaves hold one primitive each): that test comes now.  (Closest hit on an
                     * analytic primitive: after the hit, below - a rectangle can report a hit in front of its own bounds.) */
                    float te;
                    if (COUNT) rn += 1;
                    if (!slab_test6(pr.bmin[0], pr.bmin[1], pr.bmin[2], pr.bmax[0], pr.bmax[1], pr.bmax[2], ro, inv, tmax, &te)) continue;
                }
                if (geom) {
                    cp = p;
                    if (PAIRS == PT_PAIRS_MIXED) in_cloud = pr.type == SH_CLOUD;
                    node = pr.root;
                    PT_PUSH(GEOM_BOTTOM); /* under the mesh's entries: popping it leaves the mesh */
                    mode = TM_GEOM;
                } else {
                    float t;
                    int face;
                    if (COUNT) ri += 1;
#ifdef PT_STATS
                    const long long cp0_ = clock64();
                    PS_WAVE_IT(PS_TOPPRIM_IT, PS_TOPPRIM_LANES);
#endif
                    const bool ph_ = shape_hit_t<PROC>(sc, pr, ro, rd, tmax, &t, &face);
#ifdef PT_STATS
                    st_[PS_CYC_TOPPRIM] += clock64() - cp0_;
#endif
                    if (ph_) {
                        if (PT_FZ_CLOSEST) {
                            float te = 0;
                            /* a ray that misses the bounds altogether never reaches this primitive in the reference: no candidate */
                            if (!slab_test6(pr.bmin[0], pr.bmin[1], pr.bmin[2], pr.bmax[0], pr.bmax[1], pr.bmax[2], ro, inv, kInfinity, &te)) continue;
                            const float g = fmaxr(te, t * FZ_WIDEN);
                            /* a hit the reference would only take with a larger tMax than the bound of this walk (te > tmax), a hit
                             * behind the best one, a best one inside the new bound: a second candidate */
                            if (!(te <= tmax) || !(g >= t) || (hit && (t > tbest || tbest < g))) {
                                PT_DEFER();
                                break;
                            }
                            tbest = t;
                            tmax = g;
                        } else tmax = t;
                        hp = p; hf = face; if (Source::BARY) { hb0 = 0; hb1 = 0; } hit = true;
                        if (PT_ANY) { PT_FINISH(); break; }
                    }
                    if (TOP_FUSE && leaf_pos == tnode) { /* the leaf holds nothing more: the pop now, not in a step of its own */
                        tnode = PT_POP();
                        fresh = false;
                        mode = TM_TOP;
                    }
                }
            }
            if (TOP_FUSE && mode == TM_TOP && tnode == -1) { /* ... and the end of the walk in the step whose pop brought it up */
                PT_TOP_END();
                break;
            }
        }

        /* ---- geometry level: inner nodes until a leaf ---- */
        const long long c2_ = PS_CLOCK();
#if PT_SPEC
#define PT_IN_ACTIVE (mode == TM_GEOM && (node >= 0 || (pleaf == 0 && node != GEOM_BOTTOM)))
#define PT_NOT_POSTPONED (pleaf == 0)
#else
#define PT_IN_ACTIVE (mode == TM_GEOM && node >= 0)
#define PT_NOT_POSTPONED true
#endif
        while (PT_IN_ACTIVE) {
            PS_WAVE_IT(PS_IN_IT, PS_IN_LANES);
#if PT_SPEC
            if (node < 0) { /* a leaf, and none kept aside yet: keep it and take the next entry */
                pleaf = node;
                node = PT_POP();
            }
            if (node >= 0) {
#endif
            int r0, r1, r2, r3;
            const int nh = PT_VISIT(node, r0, r1, r2, r3);
#if PT_PREFETCH
            pf_acc ^= pf_pending;
            if (nh > 1) { /* touch the line the pop after this subtree will want: the nearest sibling's node or its first leaf item */
                const int *pp = r1 >= 0 ? reinterpret_cast<const int *>(sc.nodes + r1)
                                        : reinterpret_cast<const int *>(sc.items + ((unsigned)(-(r1 + 1)) & 0x07FFFFFFu));
                pf_pending = __builtin_nontemporal_load(pp);
            }
#endif
            if (nh == 0) {
                node = PT_POP();
                if (node == GEOM_BOTTOM && PT_NOT_POSTPONED) PT_LEAVE_GEOM(); /* (with a leaf kept aside: after the leaf) */
            } else {
                if (sp + 3 <= LDS_STACK) { /* the usual case: three unconditional LDS stores, farthest child deepest */
                    lds[sp * TRACE_TPB] = nh == 4 ? r3 : (nh == 3 ? r2 : r1);
                    lds[(sp + 1) * TRACE_TPB] = nh == 4 ? r2 : r1;
                    lds[(sp + 2) * TRACE_TPB] = r1;
                    sp += nh - 1;
                } else {
                    if (nh > 3) PT_PUSH(r3);
                    if (nh > 2) PT_PUSH(r2);
                    if (nh > 1) PT_PUSH(r1);
                }
                node = r0;
            }
#if PT_SPEC
            }
#endif
            if (__popcll(__ballot(PT_IN_ACTIVE)) < (TOP_BATCH > PT_TOP_BATCH ? PT_MIN_ACTIVE_WIDE : PT_MIN_ACTIVE)) break;
        }

        /* ---- leaf ---- */
        const long long c3_ = PS_CLOCK();
        if constexpr (PAIRS) {
            /* Pair mode (triangles; PAIRS == PT_PAIRS_MIXED: the leaves of particle clouds take the item loop below).  A leaf holds 1-4 triangles and a
             * quarter of the wavefront's lanes bring one per round, so the item loop below runs four times for ~15 lanes.  Here the
             * (ray, triangle) pairs of the round are dealt out to ALL lanes: lane w takes pair w, fetches the ray's constants from
             * the lane that owns it, and makes the part of the test that does not depend on tMax (tri_cand); then every owner goes
             * through its leaf's triangles in their order and accepts each against the tMax it has by then (tri_cand_accept) -
             * the reference's sequence of tests, with the same values.  One round trip to the leaf records instead of up to four. */
            for (int round_ = 0;; round_++) {
                const bool want = mode == TM_GEOM && node < 0 && !(PAIRS == PT_PAIRS_MIXED && in_cloud);
                const unsigned long long wm = __ballot(want);
                if (!wm) break;
                const unsigned packed = want ? (unsigned)(-(node + 1)) : 0u;
                const int nm1 = (int)(packed >> 27); /* triangles in the leaf - 1 */
                const unsigned long long c1m = __ballot(want && (nm1 & 1)), c2m = __ballot(want && (nm1 & 2));
                /* another round at once (for the owners the first had no place for, and for those whose next stack entry is a leaf again)
                 * only if it would be this full */
                if (round_ > 0 && (PT_LEAF_REPEAT == 0 || __popcll(wm) + __popcll(c1m) + 2 * __popcll(c2m) < PT_LEAF_REPEAT)) break;
                const int pre = __popcll(wm & lt_mask) + __popcll(c1m & lt_mask) + 2 * __popcll(c2m & lt_mask);
                /* a round serves the owners whose pairs fit its 64 places (lowest lanes first); the others keep their leaf for the next */
                const bool served = want && pre + nm1 < 64;
                const unsigned long long sm = __ballot(served);
                const int total = __popcll(sm) + __popcll(c1m & sm) + 2 * __popcll(c2m & sm);
                if (served) {
                    pairs[pre] = (unsigned char)lane;
                    if (nm1 >= 1) pairs[pre + 1] = (unsigned char)(lane | 64);
                    if (nm1 >= 2) pairs[pre + 2] = (unsigned char)(lane | 128);
                    if (nm1 >= 3) pairs[pre + 3] = (unsigned char)(lane | 192);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int dsc = pairs[lane];
                const int oaddr = (dsc & 63) << 2;
                /* the owner's ray: origin, shear, and the leaf's first record with the permutation's kz above it */
                const int w_first = __builtin_amdgcn_ds_bpermute(oaddr, (int)(packed & 0x07FFFFFFu) | (tr.kz << 28));
                f3 w_ro;
                TriRay w_tr;
                w_ro.x = __int_as_float(__builtin_amdgcn_ds_bpermute(oaddr, __float_as_int(ro.x)));
                w_ro.y = __int_as_float(__builtin_amdgcn_ds_bpermute(oaddr, __float_as_int(ro.y)));
                w_ro.z = __int_as_float(__builtin_amdgcn_ds_bpermute(oaddr, __float_as_int(ro.z)));
                w_tr.Sx = __int_as_float(__builtin_amdgcn_ds_bpermute(oaddr, __float_as_int(tr.Sx)));
                w_tr.Sy = __int_as_float(__builtin_amdgcn_ds_bpermute(oaddr, __float_as_int(tr.Sy)));
                w_tr.Sz = __int_as_float(__builtin_amdgcn_ds_bpermute(oaddr, __float_as_int(tr.Sz)));
                float c_t = 0.f, c_ts = 0.f, c_det = 0.f, c_b0 = 0.f, c_b1 = 0.f;
                int c_face = -1;
                if (lane < total) {
                    PS_WAVE_IT(PS_LEAF_IT, PS_LEAF_LANES);
                    w_tr.kz = (int)((unsigned)w_first >> 28);
                    w_tr.kx = w_tr.kz + 1; if (w_tr.kx == 3) w_tr.kx = 0;
                    w_tr.ky = w_tr.kx + 1; if (w_tr.ky == 3) w_tr.ky = 0;
                    const float4 *ip = reinterpret_cast<const float4 *>(sc.items + (w_first & 0x07FFFFFF) + (dsc >> 6));
                    const float4 q0 = ip[0], q1 = ip[1], q2 = ip[2];
                    tri_cand(w_ro, w_tr, f3(q0.x, q0.y, q0.z), f3(q1.x, q1.y, q1.z), f3(q2.x, q2.y, q2.z), &c_t, &c_ts, &c_det, &c_b0, &c_b1);
                    c_face = __float_as_int(q1.w);
                }
                bool ended = false, deferred = false;
                for (int i = 0; i < 4; i++) {
                    if (i > 0 && __ballot(served && nm1 >= i) == 0) break; /* (wave-uniform) */
                    const int saddr = ((pre + i) & 63) << 2;
                    const float t = __int_as_float(__builtin_amdgcn_ds_bpermute(saddr, __float_as_int(c_t)));
                    const float ts = __int_as_float(__builtin_amdgcn_ds_bpermute(saddr, __float_as_int(c_ts)));
                    const float det = __int_as_float(__builtin_amdgcn_ds_bpermute(saddr, __float_as_int(c_det)));
                    const int face = __builtin_amdgcn_ds_bpermute(saddr, c_face);
                    float b0 = 0, b1 = 0;
                    if (Source::BARY) {
                        b0 = __int_as_float(__builtin_amdgcn_ds_bpermute(saddr, __float_as_int(c_b0)));
                        b1 = __int_as_float(__builtin_amdgcn_ds_bpermute(saddr, __float_as_int(c_b1)));
                    }
                    if (served && nm1 >= i && !ended && !deferred) {
                        if (COUNT) ri += 1;
                        if (tri_cand_accept(ts, det, tmax)) {
                            /* FAST closest hit: the first item accepted after a hit on ANOTHER primitive is a second candidate */
                            if (PT_FZ_CLOSEST && hit && hp != cp && (t > tbest || tbest < t * FZ_WIDEN)) deferred = true;
                            else {
                                if (PT_FZ_CLOSEST) tbest = t;
                                tmax = t; hp = cp; hf = face; if (Source::BARY) { hb0 = b0; hb1 = b1; } hit = true;
                                if (PT_ANY) ended = true;
                            }
                        }
                    }
                }
                if (served) {
                    PS_WAVE_IT(PS_LEAF_ENTRY, PS_LEAF_ENTRY_LANES);
                    if (deferred) PT_DEFER();
                    else if (ended) PT_FINISH();
                    else {
                        node = PT_POP();
                        if (node == GEOM_BOTTOM) PT_LEAVE_GEOM();
                    }
                }
            }
        }
#if PT_SPEC
        if (mode == TM_GEOM && (pleaf != 0 || node < 0)) { /* (in a mesh, node == GEOM_BOTTOM only with a leaf kept aside) */
            PS_WAVE_IT(PS_LEAF_ENTRY, PS_LEAF_ENTRY_LANES);
            unsigned packed = (unsigned)(-((pleaf != 0 ? pleaf : node) + 1));
#else
        if (PAIRS != PT_PAIRS_TRI && mode == TM_GEOM && node < 0 && (PAIRS == 0 || in_cloud)) {
            PS_WAVE_IT(PS_LEAF_ENTRY, PS_LEAF_ENTRY_LANES);
            unsigned packed = (unsigned)(-(node + 1));
#endif
            unsigned first = packed & 0x07FFFFFFu, n = (packed >> 27) + 1;
            bool ended = false, deferred = false;
            /* FAST closest hit: the first item accepted after a hit on ANOTHER primitive is a second candidate (inside one
             * primitive the walk is the reference's own, whatever tree led here) */
#define PT_FZ_ACCEPT_ITEM(t_)                                                                             \
    if (PT_FZ_CLOSEST) {                                                                                  \
        if (hit && hp != cp && ((t_) > tbest || tbest < (t_) * FZ_WIDEN)) { deferred = true; break; }     \
        tbest = (t_);                                                                                     \
    }
            for (unsigned i = 0; i < n; i++) {
                PS_WAVE_IT(PS_LEAF_IT, PS_LEAF_LANES);
                const float4 *ip = reinterpret_cast<const float4 *>(sc.items + first + i);
                float4 q0 = ip[0], q1 = ip[1], q2 = ip[2];
                int kind = __float_as_int(q2.w);
                if (COUNT) ri += 1;
                if (kind == ITEM_TRI) {
                    float t, b0, b1, b2;
                    if (tri_hit_pre(ro, tr, tmax, f3(q0.x, q0.y, q0.z), f3(q1.x, q1.y, q1.z), f3(q2.x, q2.y, q2.z), &t, &b0, &b1, &b2)) {
                        PT_FZ_ACCEPT_ITEM(t)
                        tmax = t; hp = cp; hf = __float_as_int(q1.w); if (Source::BARY) { hb0 = b0; hb1 = b1; } hit = true;
                        if (PT_ANY) { ended = true; break; }
                    }
                } else {
                    f3 o, dd;
                    particle_object_ray(ro, rd, f3(q0.x, q0.y, q0.z), &o, &dd);
                    float t;
                    if (particle_box_test(ro, inv, tmax, f3(q0.x, q0.y, q0.z), q1.x) && sphere_hit_t_obj(o, dd, q1.x, tmax, &t)) {
                        PT_FZ_ACCEPT_ITEM(t)
                        tmax = t; hp = cp; hf = __float_as_int(q1.w); if (Source::BARY) { hb0 = 0; hb1 = 0; } hit = true;
                        if (PT_ANY) { ended = true; break; }
                    }
                }
            }
#undef PT_FZ_ACCEPT_ITEM
            if (deferred) PT_DEFER();
            else if (ended) PT_FINISH();
#if PT_SPEC
            else if (pleaf != 0) { /* the leaf kept aside is done; `node` is what the lane walked on to: an inner node, the next leaf, or the end */
                if (node == GEOM_BOTTOM) PT_LEAVE_GEOM();
            }
#endif
            else {
                node = PT_POP();
                if (node == GEOM_BOTTOM) PT_LEAVE_GEOM();
            }
#if PT_SPEC
            pleaf = 0;
#endif
        }
#ifdef PT_STATS
        const long long c4_ = clock64();
        st_[PS_CYC_REFILL] += c1_ - c0_; st_[PS_CYC_TOP] += c2_ - c1_; st_[PS_CYC_IN] += c3_ - c2_; st_[PS_CYC_LEAF] += c4_ - c3_;
#endif
    }
#ifdef PT_STATS
    st_[PS_CYC_TOTAL] = clock64() - t_begin_;
    if (lane == 0) {
        const int ks[] = {PS_OUTER, PS_CYC_REFILL, PS_CYC_TOP, PS_CYC_IN, PS_CYC_LEAF, PS_CYC_TOTAL, PS_CYC_TOPPRIM};
        for (int k : ks) atomicAdd(&g_pt_stats[k], st_[k]);
    }
#endif
#if PT_PREFETCH
    if (pf_acc == 0x5a5a5a5a && n_rays < 0) *total_nodes = (unsigned)pf_acc; /* keeps the touches alive; never taken */
#endif
#undef PT_FINISH
#undef PT_ANY
#undef PT_FZ_CLOSEST
#undef PT_DEFER
#undef PT_LEAVE_GEOM
#undef PT_TOP_END
}

/* ---- fullest-phase-first scheduling -------------------------------------------------------------------
 * persistent_trace above runs its phases one after the other in every round - top-level steps, inner nodes until
 * few lanes are left in them, then the leaf items - each for the lanes that happen to be in it.  Measured on
 * config[1] (tools/gpu_tracestats.py, PT_STATS build, r02): 44 / 26 / 24 % of the lanes are active in a top /
 * inner / leaf iteration; a leaf phase starts with 19 of 64 lanes and loops over 1..4 items.  With lanes spread
 * over four states and every state served once per round, a quarter of the lanes per phase is what that schedule gives.
 *
 * Here a round serves ONE phase: the one most lanes wait for (weighted by PT_W_*).  Lanes in the other states
 * keep their state and wait; whatever is not served fills up until it is the fullest, so nothing starves, and
 * a step runs with the largest group the wavefront has.  A step is one top-level iteration, one node visit, or
 * ONE leaf item (the item loop is part of the state: no 1..4 trip count inside a step).  Same per-ray
 * operations, in the same order per ray, as persistent_trace. */
enum { TS_IDLE = 0, TS_TOP = 1, TS_TOPLEAF = 2, TS_IN = 3, TS_LEAF = 4 };
#ifndef PT_W_TOP
#define PT_W_TOP 4
#endif
#ifndef PT_W_IN
#define PT_W_IN 4
#endif
#ifndef PT_W_LEAF
#define PT_W_LEAF 4
#endif
#ifndef PT_W_IDLE
#define PT_W_IDLE 4
#endif
/* hysteresis: the phase served last is served again while at least this many lanes are in it (0: always the fullest) */
#ifndef PT_HYST
#define PT_HYST 0
#endif
template <class Source, bool COUNT>
PD void persistent_trace_sched(const DScene &sc, Source src, int *lds, const float4 *staged, unsigned *total_nodes, unsigned *total_items) {
    int ovf[OVF_STACK];
    const int lane = __lane_id();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#ifdef PT_STATS
    unsigned long long st_[PS_N] = {};
    const long long t_begin_ = clock64();
#endif

    int mode = TS_IDLE, slot = -1, tag = 0;
    src.init();
    f3 ro, rd, inv;
    TriRay tr;
    tr.kx = tr.ky = tr.kz = 0; tr.Sx = tr.Sy = tr.Sz = 0;
    float tmax = 0, hb0 = 0, hb1 = 0;
    int hp = -1, hf = -1;
    bool hit = false;
    int tnode = -1, leaf_pos = 0, leaf_end = 0, cp = -1, node = 0, sp = 0, sp_base = 0;
    unsigned it_pos = 0, it_end = 0; /* TS_LEAF: items left in the leaf */
    unsigned rn = 0, ri = 0;

    const int n_rays = src.total();
    int pool_pos = 0, pool_end = 0; /* wave-uniform */
    bool exhausted = false;
    int last_pick = TS_IDLE; /* wave-uniform */

#define PT_FINISH()                                                                  \
    do {                                                                             \
        src.template store<COUNT>(tag, slot, hit, hp, hf, tmax, hb0, hb1, rn, ri);   \
        if (COUNT) { *total_nodes += rn; *total_items += ri; }                       \
        mode = TS_IDLE;                                                              \
        slot = -1;                                                                   \
    } while (0)
#define PT_ANY (src.is_any(tag))
/* a geometry ref taken from a node or the stack: an inner node, or a leaf whose items become the lane's state */
#define PT_ENTER(ref)                                                                \
    do {                                                                             \
        node = (ref);                                                                \
        if (node >= 0) mode = TS_IN;                                                 \
        else {                                                                       \
            const unsigned packed_ = (unsigned)(-(node + 1));                        \
            it_pos = packed_ & 0x07FFFFFFu;                                          \
            it_end = it_pos + (packed_ >> 27) + 1;                                   \
            mode = TS_LEAF;                                                          \
        }                                                                            \
    } while (0)
/* the geometry BVH of the current primitive has nothing left for this ray: next primitive of the top-level leaf */
#define PT_GEOM_NEXT()                                                               \
    do {                                                                             \
        if (sp == sp_base) mode = TS_TOPLEAF;                                        \
        else { const int r_ = PT_POP(); PT_ENTER(r_); }                              \
    } while (0)

    for (;;) {
        PS_ADD(PS_OUTER, 1);
        const long long c0_ = PS_CLOCK();
        const int n_idle = __popcll(__ballot(mode == TS_IDLE));
        const int n_top = __popcll(__ballot(mode == TS_TOP || mode == TS_TOPLEAF));
        const int n_in = __popcll(__ballot(mode == TS_IN));
        const int n_leaf = __popcll(__ballot(mode == TS_LEAF));
        const int w_top = n_top * PT_W_TOP, w_in = n_in * PT_W_IN, w_leaf = n_leaf * PT_W_LEAF;
        const int w_busy = w_top > w_in ? (w_top > w_leaf ? w_top : w_leaf) : (w_in > w_leaf ? w_in : w_leaf);
        if (n_idle == 64 && exhausted) break; /* every wave reaches this: the grid drains */

        if (!exhausted && n_idle > 0 && n_idle * PT_W_IDLE >= w_busy) {
            /* ---- refill idle lanes ---- */
            if (pool_pos >= pool_end) {
                int base = 0;
                if (lane == 0) base = atomicAdd(src.counter, PT_POOL);
                base = __shfl(base, 0, 64);
                if (base >= n_rays) exhausted = true;
                else {
                    pool_pos = base;
                    pool_end = base + PT_POOL < n_rays ? base + PT_POOL : n_rays;
                }
            }
            const int avail = pool_end - pool_pos;
            if (avail > 0) {
                const unsigned long long idle = __ballot(mode == TS_IDLE);
                const int rank = __popcll(idle & lt_mask);
                if (mode == TS_IDLE && rank < avail) {
                    PS_WAVE_IT(PS_REFILL_IT, PS_REFILL_LANES);
                    const int idx = pool_pos + rank;
                    src.fetch(idx, &tag, &slot, &ro, &rd, &tmax);
                    inv = f3(1 / rd.x, 1 / rd.y, 1 / rd.z);
                    tr = tri_ray_setup(rd);
                    hp = -1; hf = -1; hb0 = 0; hb1 = 0; hit = false;
                    rn = 0; ri = 0;
                    sp = 0;
                    if (sc.n_top == 0) PT_FINISH();
                    else {
                        PT_PUSH(-1);
                        tnode = 0;
                        mode = TS_TOP;
                    }
                }
                pool_pos += n_idle < avail ? n_idle : avail;
            }
#ifdef PT_STATS
            st_[PS_CYC_REFILL] += clock64() - c0_;
#endif
            continue;
        }

        int pick = (w_in >= w_leaf && w_in >= w_top) ? TS_IN : (w_leaf >= w_top ? TS_LEAF : TS_TOP);
        if (PT_HYST > 0) {
            const int n_last = last_pick == TS_IN ? n_in : (last_pick == TS_LEAF ? n_leaf : (last_pick == TS_TOP ? n_top : 0));
            if (n_last >= PT_HYST) pick = last_pick;
            last_pick = pick;
        }
        if (pick == TS_IN) {
            /* ---- one node visit of the geometry BVH ---- */
            if (mode == TS_IN) {
                PS_WAVE_IT(PS_IN_IT, PS_IN_LANES);
                int r0, r1, r2, r3;
                const int nh = PT_VISIT(node, r0, r1, r2, r3);
                if (nh == 0) PT_GEOM_NEXT();
                else {
                    if (sp + 3 <= LDS_STACK) { /* the usual case: three unconditional LDS stores, farthest child deepest */
                        lds[sp * TRACE_TPB] = nh == 4 ? r3 : (nh == 3 ? r2 : r1);
                        lds[(sp + 1) * TRACE_TPB] = nh == 4 ? r2 : r1;
                        lds[(sp + 2) * TRACE_TPB] = r1;
                        sp += nh - 1;
                    } else {
                        if (nh > 3) PT_PUSH(r3);
                        if (nh > 2) PT_PUSH(r2);
                        if (nh > 1) PT_PUSH(r1);
                    }
                    PT_ENTER(r0);
                }
            }
#ifdef PT_STATS
            st_[PS_CYC_IN] += clock64() - c0_;
#endif
        } else if (pick == TS_LEAF) {
            /* ---- one leaf item ---- */
            if (mode == TS_LEAF) {
                PS_WAVE_IT(PS_LEAF_IT, PS_LEAF_LANES);
                const float4 *ip = reinterpret_cast<const float4 *>(sc.items + it_pos);
                const float4 q0 = ip[0], q1 = ip[1], q2 = ip[2];
                const int kind = __float_as_int(q2.w);
                it_pos++;
                if (COUNT) ri += 1;
                bool ended = false;
                if (kind == ITEM_TRI) {
                    float t, b0, b1, b2;
                    if (tri_hit_pre(ro, tr, tmax, f3(q0.x, q0.y, q0.z), f3(q1.x, q1.y, q1.z), f3(q2.x, q2.y, q2.z), &t, &b0, &b1, &b2)) {
                        tmax = t; hp = cp; hf = __float_as_int(q1.w); hb0 = b0; hb1 = b1; hit = true;
                        ended = PT_ANY;
                    }
                } else {
                    f3 o, dd;
                    particle_object_ray(ro, rd, f3(q0.x, q0.y, q0.z), &o, &dd);
                    float t;
                    if (particle_box_test(ro, inv, tmax, f3(q0.x, q0.y, q0.z), q1.x) && sphere_hit_t_obj(o, dd, q1.x, tmax, &t)) {
                        tmax = t; hp = cp; hf = __float_as_int(q1.w); hb0 = 0; hb1 = 0; hit = true;
                        ended = PT_ANY;
                    }
                }
                if (ended) PT_FINISH();
                else if (it_pos == it_end) PT_GEOM_NEXT();
            }
#ifdef PT_STATS
            st_[PS_CYC_LEAF] += clock64() - c0_;
#endif
        } else {
            /* ---- one step of the top level: the reference's own tree, step for step (primitive.cpp:374-431) ---- */
            if (mode == TS_TOP || mode == TS_TOPLEAF) {
                PS_WAVE_IT(PS_TOP_IT, PS_TOP_LANES);
                if (mode == TS_TOPLEAF && leaf_pos == leaf_end) { /* leaf done: the popped node is tested in this same step */
                    tnode = PT_POP();
                    mode = TS_TOP;
                }
                if (mode == TS_TOP) {
                    if (tnode == -1) PT_FINISH();
                    else {
                        const TopNode &nd = sc.top[tnode];
                        float te;
                        if (COUNT) rn += 1;
                        if (slab_test(nd.bmin, nd.bmax, ro, inv, tmax, &te)) {
                            if (nd.left < 0) {
                                leaf_pos = nd.first;
                                leaf_end = nd.first + nd.n;
                                mode = TS_TOPLEAF;
                            } else {
                                const TopNode &l = sc.top[nd.left];
                                const TopNode &r = sc.top[nd.right];
                                float tl0, tr0;
                                if (COUNT) rn += 2;
                                bool visitL = slab_test(l.bmin, l.bmax, ro, inv, tmax, &tl0);
                                bool visitR = slab_test(r.bmin, r.bmax, ro, inv, tmax, &tr0);
                                if (visitR && visitL) {
                                    int first, second;
                                    if (tr0 < tl0) { first = nd.right; second = nd.left; }
                                    else { first = nd.left; second = nd.right; }
                                    PT_PUSH(second);
                                    tnode = first;
                                } else if (visitL) tnode = nd.left;
                                else if (visitR) tnode = nd.right;
                                else tnode = PT_POP();
                            }
                        } else tnode = PT_POP();
                        if (mode == TS_TOP && tnode == -1) PT_FINISH(); /* nothing left: no extra round for the sentinel */
                    }
                } else {
                    const int p = sc.order[leaf_pos++];
                    const DPrim &pr = sc.prims[p];
                    if (pr.type == SH_MESH || pr.type == SH_CLOUD) {
                        cp = p;
                        sp_base = sp;
                        PT_ENTER(pr.root);
                    } else {
                        float t;
                        int face;
                        if (COUNT) ri += 1;
                        if (shape_hit_t(sc, pr, ro, rd, tmax, &t, &face)) {
                            tmax = t; hp = p; hf = face; hb0 = 0; hb1 = 0; hit = true;
                            if (PT_ANY) PT_FINISH();
                        }
                    }
                }
            }
#ifdef PT_STATS
            st_[PS_CYC_TOP] += clock64() - c0_;
#endif
        }
    }
#ifdef PT_STATS
    st_[PS_CYC_TOTAL] = clock64() - t_begin_;
    if (lane == 0) {
        const int ks[] = {PS_OUTER, PS_CYC_REFILL, PS_CYC_TOP, PS_CYC_IN, PS_CYC_LEAF, PS_CYC_TOTAL};
        for (int k : ks) atomicAdd(&g_pt_stats[k], st_[k]);
    }
#endif
#undef PT_FINISH
#undef PT_ANY
#undef PT_ENTER
#undef PT_GEOM_NEXT
}

/* which schedule the kernels of rt_trace.hip use */
#ifndef PT_SCHED
#define PT_SCHED 0 /* r02 on config[1]: phases in sequence 190.5 ms per 32-spp pass, fullest-phase-first 193.2 */
#endif
template <class Source, bool COUNT, bool FAST = false, int TOP_STAGED = TOP_LDS_NODES, int TOP_BATCH = PT_TOP_BATCH, bool PROC = true, int PAIRS = 0>
PD void persistent_trace_any(const DScene &sc, Source src, int *lds, const float4 *staged, const float4 *top_staged, unsigned *total_nodes,
                             unsigned *total_items, unsigned char *pairs = nullptr) {
#if PT_SCHED
    static_assert(!FAST || !PT_SCHED, "the fullest-phase-first variant walks the reference's tree only");
    persistent_trace_sched<Source, COUNT>(sc, src, lds, staged, total_nodes, total_items);
#else
    persistent_trace<Source, COUNT, FAST, TOP_STAGED, TOP_BATCH, PROC, PAIRS>(sc, src, lds, staged, top_staged, total_nodes, total_items, pairs);
#endif
}
