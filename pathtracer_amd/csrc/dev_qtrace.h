/*
 * dev_qtrace.h — the quad tracer: persistent wavefronts in which FOUR lanes share one ray.
 *
 * Same scene query as dev_trace.h (Aggregator::Intersect, primitive.cpp:374-431, over the reference's own
 * top-level tree, then our 4-wide geometry BVH) and the same per-ray arithmetic, so ids, t and barycentrics
 * are those of the one-ray-per-lane tracer bit for bit.  What changes is who does the work:
 *
 *   node visit   lane k of a quad fetches child k of the node (two 16-byte loads: Node128 is child-major) and
 *                runs ONE slab test; the four (entry distance, ref) pairs are exchanged inside the quad with
 *                DPP quad_perm moves (no LDS, no memory) and every lane runs the same 5-comparator network the
 *                scalar walker uses, so the visiting order is unchanged.  The up-to-three far children are pushed
 *                with ONE ds_write: lane k stores entry k.
 *   leaf         lane k tests item k of the (<= 4) items: one watertight triangle test or one particle
 *                quadratic per lane instead of a loop of up to four; the quad's closest accepted t comes
 *                from two DPP min steps.  The hit record (face, barycentrics) stays in the lane that
 *                found it (`have`), which is the lane that stores the result when the ray ends.
 *   stack        one per quad, wholly in LDS (QT_STACK entries; the scalar tracer keeps 24 per lane in LDS and
 *                the rest in scratch): no scratch memory at all in this kernel.
 *
 * Why: measured on config[1] (profiles/r01k), the one-ray-per-lane state machine issues ~156 wave
 * instructions per ray at 24 % of its lanes active - the three phases (top level, inner nodes, leaf items)
 * run one after the other, each for the lanes that happen to be in it, and a leaf is a loop of 1..4 items.
 * With a quad per ray a node visit is ~1/3 of the instructions, a leaf is one test deep whatever its
 * size, and 16 rays (not 64) have to agree on a phase.
 *
 * Sequential semantics kept: the reference accepts t <= tMax item after item, so of the items of a leaf
 * the one with the smallest t wins and, at bit-equal t, the one tested last: here the highest lane.
 * Top-level steps (tens of nodes; analytic primitives) are executed by all four lanes alike.
 */
#pragma once
#include "dev_trace.h"

#ifndef QT_STACK
#define QT_STACK 96 /* entries per quad, all in LDS: (QT_STACK * QT_STRIDE) ints per 256-thread block */
#endif
#define QT_QUADS (TRACE_TPB / 4)
#define QT_STRIDE (QT_QUADS + 1) /* entry e of quad q at e * 65 + q: the three pushes of a visit fall into three banks */
#define QT_LDS_INTS (QT_STACK * QT_STRIDE)
/* the inner-node loop hands control back once fewer than this many quads are in it */
#ifndef QT_MIN_ACTIVE
#define QT_MIN_ACTIVE 2
#endif
/* quads that want a top-level step wait until this many do */
#ifndef QT_TOP_BATCH
#define QT_TOP_BATCH 4
#endif
/* idle quads wait until this many can be refilled together */
#ifndef QT_REFILL
#define QT_REFILL 6
#endif

template <int CTRL> PD int qt_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> PD float qt_dpp_f(float v) { return __int_as_float(qt_dpp_i<CTRL>(__float_as_int(v))); }
/* quad_perm controls: broadcast of lane k = k * 0x55; xor 1 = [1,0,3,2]; xor 2 = [2,3,0,1] */
#define QT_BCAST(k) ((k) * 0x55)
#define QT_XOR1 0xB1
#define QT_XOR2 0x4E
PD float qt_quad_min(float v) {
    v = fminf(v, qt_dpp_f<QT_XOR1>(v));
    return fminf(v, qt_dpp_f<QT_XOR2>(v));
}
PD unsigned qt_quad_sum(unsigned v) {
    v += (unsigned)qt_dpp_i<QT_XOR1>((int)v);
    return v + (unsigned)qt_dpp_i<QT_XOR2>((int)v);
}

#define QT_PUSH(v)                                                   \
    do {                                                             \
        if (sub == 0 && sp < QT_STACK) lds[sp * QT_STRIDE] = (v);    \
        sp++;                                                        \
    } while (0)
#define QT_POP() (--sp, lds[(sp < QT_STACK ? sp : QT_STACK - 1) * QT_STRIDE])

template <class Source, bool COUNT>
PD void quad_trace(const DScene &sc, Source src, int *lds /* &stack[quad of this lane] */, unsigned *total_nodes, unsigned *total_items) {
    const int lane = __lane_id();
    const int sub = lane & 3;
    const int qbit = lane & ~3; /* first lane of this quad */
    const unsigned long long first_lanes = 0x1111111111111111ull;
    const unsigned long long quads_before = ((1ull << qbit) - 1ull) & first_lanes;

    int mode = TM_IDLE, slot = -1, tag = 0;
    src.init();
    f3 ro, rd, inv;
    TriRay tr;
    tr.kx = tr.ky = tr.kz = 0; tr.Sx = tr.Sy = tr.Sz = 0;
    float tmax = 0, hb0 = 0, hb1 = 0;
    int hp = -1, hf = -1;
    bool hit = false;  /* the ray has an accepted hit (same in the four lanes) */
    bool have = false; /* ... and this lane holds its record */
    int tnode = -1, leaf_pos = 0, leaf_end = 0, cp = -1, node = 0, sp = 0, sp_base = 0;
    unsigned rn = 0, ri = 0; /* per-lane work (COUNT), summed over the quad when the ray ends */

    const int n_rays = src.total();
    int pool_pos = 0, pool_end = 0; /* wave-uniform */
    bool exhausted = false;

#define QT_FINISH()                                                                               \
    do {                                                                                          \
        if (COUNT) { rn = qt_quad_sum(rn); ri = qt_quad_sum(ri); }                                \
        if (hit ? have : sub == 0) src.template store<COUNT>(tag, slot, hit, hp, hf, tmax, hb0, hb1, rn, ri); \
        if (COUNT && sub == 0) { *total_nodes += rn; *total_items += ri; }                        \
        mode = TM_IDLE;                                                                           \
        slot = -1;                                                                                \
    } while (0)
#define QT_ANY (src.is_any(tag))

    for (;;) {
        /* ---- refill idle quads ---- */
        const unsigned long long idle = __ballot(mode == TM_IDLE) & first_lanes;
        if (idle && (__popcll(idle) >= QT_REFILL || __ballot(mode == TM_GEOM) == 0)) {
            if (pool_pos >= pool_end && !exhausted) {
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
                const int rank = __popcll(idle & quads_before);
                if (mode == TM_IDLE && rank < avail) {
                    const int idx = pool_pos + rank;
                    src.fetch(idx, &tag, &slot, &ro, &rd, &tmax);
                    inv = f3(1 / rd.x, 1 / rd.y, 1 / rd.z);
                    tr = tri_ray_setup(rd);
                    hp = -1; hf = -1; hb0 = 0; hb1 = 0; hit = false; have = false;
                    rn = 0; ri = 0;
                    sp = 0;
                    if (sc.n_top == 0) QT_FINISH();
                    else {
                        QT_PUSH(-1);
                        tnode = 0;
                        mode = TM_TOP;
                    }
                }
                const int want = __popcll(idle);
                pool_pos += want < avail ? want : avail;
            }
            if (exhausted && __ballot(mode != TM_IDLE) == 0) break; /* every wave reaches this: the grid drains */
        }

        /* ---- top level: the reference's own tree, step for step (primitive.cpp:374-431); the four lanes of a
         * quad run it alike ---- */
        const unsigned long long in_top = __ballot(mode == TM_TOP || mode == TM_TOPLEAF) & first_lanes;
        if (__popcll(in_top) >= QT_TOP_BATCH || __ballot(mode == TM_GEOM) == 0)
        while (mode == TM_TOP || mode == TM_TOPLEAF) {
            if (mode == TM_TOP) {
                if (tnode == -1) { QT_FINISH(); break; }
                const TopNode &nd = sc.top[tnode];
                float te;
                if (COUNT) rn += sub == 0;
                if (slab_test(nd.bmin, nd.bmax, ro, inv, tmax, &te)) {
                    if (nd.left < 0) {
                        leaf_pos = nd.first;
                        leaf_end = nd.first + nd.n;
                        mode = TM_TOPLEAF;
                    } else {
                        const TopNode &l = sc.top[nd.left];
                        const TopNode &r = sc.top[nd.right];
                        float tl0, tr0;
                        if (COUNT) rn += sub == 0 ? 2 : 0;
                        bool visitL = slab_test(l.bmin, l.bmax, ro, inv, tmax, &tl0);
                        bool visitR = slab_test(r.bmin, r.bmax, ro, inv, tmax, &tr0);
                        if (visitR && visitL) {
                            int first, second;
                            if (tr0 < tl0) { first = nd.right; second = nd.left; }
                            else { first = nd.left; second = nd.right; }
                            QT_PUSH(second);
                            tnode = first;
                        } else if (visitL) tnode = nd.left;
                        else if (visitR) tnode = nd.right;
                        else tnode = QT_POP();
                    }
                } else tnode = QT_POP();
            } else {
                if (leaf_pos == leaf_end) {
                    tnode = QT_POP();
                    mode = TM_TOP;
                } else {
                    const int p = sc.order[leaf_pos++];
                    const DPrim &pr = sc.prims[p];
                    if (pr.type == SH_MESH || pr.type == SH_CLOUD) {
                        cp = p;
                        node = pr.root;
                        sp_base = sp;
                        mode = TM_GEOM;
                    } else {
                        float t;
                        int face;
                        if (COUNT) ri += sub == 0;
                        if (shape_hit_t(sc, pr, ro, rd, tmax, &t, &face)) {
                            tmax = t; hp = p; hf = face; hb0 = 0; hb1 = 0; hit = true; have = sub == 0;
                            if (QT_ANY) { QT_FINISH(); break; }
                        }
                    }
                }
            }
        }

        /* ---- geometry level: inner nodes until a leaf; lane k of the quad takes child k ---- */
        while (mode == TM_GEOM && node >= 0) {
            const float4 *np = reinterpret_cast<const float4 *>(sc.nodes + node) + 2 * sub;
            const float4 a = np[0], b = np[1];
            int ref = __float_as_int(b.z);
            float te;
            if (COUNT) rn += ref != BVH_NO_CHILD;
            const bool h = slab_minmax(a.x, a.y, a.z, a.w, b.x, b.y, ro, inv, tmax, &te) & (ref != BVH_NO_CHILD);
            ref = h ? ref : BVH_NO_CHILD;
            te = h ? te : kInfinity;
            float t0 = qt_dpp_f<QT_BCAST(0)>(te), t1 = qt_dpp_f<QT_BCAST(1)>(te), t2 = qt_dpp_f<QT_BCAST(2)>(te), t3 = qt_dpp_f<QT_BCAST(3)>(te);
            int r0 = qt_dpp_i<QT_BCAST(0)>(ref), r1 = qt_dpp_i<QT_BCAST(1)>(ref), r2 = qt_dpp_i<QT_BCAST(2)>(ref), r3 = qt_dpp_i<QT_BCAST(3)>(ref);
            BVH4_CSWAP(t0, r0, t1, r1);
            BVH4_CSWAP(t2, r2, t3, r3);
            BVH4_CSWAP(t0, r0, t2, r2);
            BVH4_CSWAP(t1, r1, t3, r3);
            BVH4_CSWAP(t1, r1, t2, r2);
            const int nh = (r0 != BVH_NO_CHILD) + (r1 != BVH_NO_CHILD) + (r2 != BVH_NO_CHILD) + (r3 != BVH_NO_CHILD);
            if (nh == 0) {
                if (sp == sp_base) mode = TM_TOPLEAF;
                else node = QT_POP();
            } else {
                /* farthest child deepest: sorted entry k (k = 1 .. nh-1) goes to sp + nh-1-k, written by lane k */
                const int at = sp + nh - 1 - sub;
                const int mine = sub == 1 ? r1 : (sub == 2 ? r2 : r3);
                if (sub >= 1 && sub < nh && at < QT_STACK) lds[at * QT_STRIDE] = mine;
                sp += nh - 1;
                node = r0;
            }
            if (__popcll(__ballot(mode == TM_GEOM && node >= 0)) < 4 * QT_MIN_ACTIVE) break;
        }

        /* ---- leaf: lane k of the quad takes item k ---- */
        if (mode == TM_GEOM && node < 0) {
            const unsigned packed = (unsigned)(-(node + 1));
            const unsigned first = packed & 0x07FFFFFFu, n = (packed >> 27) + 1;
            const bool mine = (unsigned)sub < n;
            const float4 *ip = reinterpret_cast<const float4 *>(sc.items + first + (mine ? sub : 0));
            const float4 q0 = ip[0], q1 = ip[1], q2 = ip[2];
            const int kind = __float_as_int(q2.w);
            if (COUNT) ri += mine;
            float t = kInfinity, b0 = 0, b1 = 0, b2 = 0;
            bool h;
            if (kind == ITEM_TRI) h = tri_hit_pre(ro, tr, tmax, f3(q0.x, q0.y, q0.z), f3(q1.x, q1.y, q1.z), f3(q2.x, q2.y, q2.z), &t, &b0, &b1, &b2);
            else {
                f3 o, dd;
                particle_object_ray(ro, rd, f3(q0.x, q0.y, q0.z), &o, &dd);
                h = particle_box_test(ro, inv, tmax, f3(q0.x, q0.y, q0.z), q1.x) && sphere_hit_t_obj(o, dd, q1.x, tmax, &t);
                b0 = 0; b1 = 0;
            }
            h = h && mine;
            const unsigned hq = (unsigned)(__ballot(h) >> qbit) & 0xFu;
            bool ended = false;
            if (hq) { /* same in the four lanes */
                const float tm = qt_quad_min(h ? t : kInfinity);
                const unsigned wq = (unsigned)(__ballot(h && t == tm) >> qbit) & 0xFu; /* lanes holding the closest t: the last one tested wins */
                const int w = 31 - __clz((int)wq);
                tmax = tm;
                hit = true;
                have = sub == w;
                if (have) { hp = cp; hf = __float_as_int(q1.w); hb0 = b0; hb1 = b1; }
                if (QT_ANY) ended = true;
            }
            if (ended) QT_FINISH();
            else if (sp == sp_base) mode = TM_TOPLEAF;
            else node = QT_POP();
        }
    }
#undef QT_FINISH
#undef QT_ANY
}
