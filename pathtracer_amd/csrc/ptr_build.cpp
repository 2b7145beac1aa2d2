/*
 * ptr_build.cpp — flat scene -> device tables + unified SAH BVH (host side).
 *
 * What it replaces in the reference: MakeSceneGPU (scene.cpp:891-946: shapes, materials, media,
 * primitives), Mesh ctor (mesh.cpp:28-69: vertices to world space), Aggregator::Wrap
 * (primitive.cpp:620-642) with MeshGetBounds / ParticleCloudGetBounds / MakeSceneTable and the
 * three CreateBVH median-split builds, Aggregator::Insert / MakeLights / SetLights
 * (primitive.cpp:164-244, scene.cpp:948-958).
 *
 * Every derived scalar the render path later reads (sphere angles, world bounds, areas, the
 * light-map world radius) is computed with the reference's own expressions in float, so the
 * kernels start from bit-identical inputs; the acceleration structure itself is new.
 */
#include "ptr_host.h"
#include <chrono>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <system_error>
#include <thread>

namespace {

const double kPiD = 3.14159265358979323846;

struct V3h { float x, y, z; };
inline float radians_f(float deg) { return (float)((kPiD / 180) * deg); }
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* transform.h:135-148 on the full 4x4 (row 3 is checked to be 0,0,0,1 before) */
inline V3h xf_point16(const float *m, V3h p) {
    V3h r;
    r.x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    r.y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    r.z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    return r;
}
struct Box {
    float mn[3], mx[3];
    void reset() { for (int i = 0; i < 3; i++) { mn[i] = FLT_MAX; mx[i] = -FLT_MAX; } }
    void grow(const float *p) { for (int i = 0; i < 3; i++) { if (p[i] < mn[i]) mn[i] = p[i]; if (p[i] > mx[i]) mx[i] = p[i]; } }
    void grow(const Box &b) { /* an empty box (reset) adds nothing */
        for (int i = 0; i < 3; i++) { if (b.mn[i] < mn[i]) mn[i] = b.mn[i]; if (b.mx[i] > mx[i]) mx[i] = b.mx[i]; }
    }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0;
        return dx * dy + dy * dz + dz * dx;
    }
};
/* Transform::operator()(Bounds3f), transform.h:250-265: union of the 8 corners */
Box xf_bounds16(const float *m, const float *bmin, const float *bmax) {
    Box b;
    b.reset();
    const float *c[2] = {bmin, bmax};
    for (int k = 0; k < 8; k++) {
        V3h p = {c[k & 1][0], c[(k >> 1) & 1][1], c[(k >> 2) & 1][2]};
        V3h q = xf_point16(m, p);
        float qq[3] = {q.x, q.y, q.z};
        b.grow(qq);
    }
    return b;
}
bool is_affine(const float *m) { return m[12] == 0 && m[13] == 0 && m[14] == 0 && m[15] == 1; }
bool swaps_handedness(const float *m) { /* transform.cpp:203-208 */
    float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
    return det < 0;
}
void copy34(float *dst, const float *m16) { memcpy(dst, m16, 12 * sizeof(float)); }

DMedium conv_medium(const PtrMedium &pm) {
    DMedium m;
    for (int i = 0; i < 3; i++) {
        m.sigma_a[i] = pm.sigma_a[i];
        m.sigma_s[i] = pm.sigma_s[i];
        m.sigma_t[i] = pm.sigma_a[i] + pm.sigma_s[i];
    }
    m.g = pm.g;
    return m;
}

inline float as_float(int32_t v) { float f; memcpy(&f, &v, 4); return f; }

/* ---- binned SAH build ----------------------------------------------------------------- */
/* NodeQ (dev_scene.h) from Node128: per axis the smallest power-of-two step whose 255th multiple reaches the far side of the
 * union box, then each child plane rounded outwards AGAINST THE DECODE EXPRESSION ITSELF (fmaf(q, step, origin), what
 * dev_isect.h bvh4_visit_q computes), so that the decoded box contains the exact one whatever the rounding of that sum. */
static void quantise_nodes(const std::vector<Node128> &nodes, std::vector<NodeQ> *out) {
    out->assign(nodes.size(), NodeQ());
    for (size_t i = 0; i < nodes.size(); i++) {
        const Node128 &nd = nodes[i];
        NodeQ &q = (*out)[i];
        memset(&q, 0, sizeof(q));
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        bool any = false;
        for (int c = 0; c < 4; c++) {
            q.ref[c] = nd.c[c].ref;
            if (nd.c[c].ref == BVH_NO_CHILD) continue;
            for (int a = 0; a < 3; a++) {
                lo[a] = any ? std::min(lo[a], nd.c[c].mn[a]) : nd.c[c].mn[a];
                hi[a] = any ? std::max(hi[a], nd.c[c].mx[a]) : nd.c[c].mx[a];
            }
            any = true;
        }
        for (int a = 0; a < 3; a++) {
            q.o[a] = lo[a];
            int e = 1; /* step = 2^(e - 127), e in 1 .. 254 */
            {
                const float ext = hi[a] - lo[a];
                if (ext > 0 && std::isfinite(ext)) {
                    int ex;
                    std::frexp(ext / 255.f, &ex); /* ext / 255 = m 2^ex, m in [0.5, 1): 2^ex >= ext / 255 */
                    e = std::min(254, std::max(1, ex + 127));
                }
            }
            auto step_of = [](int e_) { uint32_t u = (uint32_t)e_ << 23; float f; memcpy(&f, &u, 4); return f; };
            while (e < 254 && !(std::fmaf(255.f, step_of(e), lo[a]) >= hi[a])) e++;
            const float st = step_of(e);
            q.e |= (uint32_t)e << (8 * a);
            for (int c = 0; c < 4; c++) {
                if (nd.c[c].ref == BVH_NO_CHILD) continue;
                int ql = (int)std::floor((nd.c[c].mn[a] - lo[a]) / st), qh = (int)std::ceil((nd.c[c].mx[a] - lo[a]) / st);
                ql = std::min(255, std::max(0, ql));
                qh = std::min(255, std::max(0, qh));
                while (ql > 0 && std::fmaf((float)ql, st, lo[a]) > nd.c[c].mn[a]) ql--;
                while (qh < 255 && std::fmaf((float)qh, st, lo[a]) < nd.c[c].mx[a]) qh++;
                q.q[a] |= (uint32_t)ql << (8 * c);
                q.q[3 + a] |= (uint32_t)qh << (8 * c);
            }
        }
    }
}

/* node of the intermediate binary SAH tree (host only; collapsed into Node128 below) */
struct Node64 {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int32_t left, right;
};

struct Builder {
    const std::vector<Box> &boxes;
    std::vector<uint32_t> idx;
    std::vector<float> cen; /* 3 per item */
    /* what a (sub)tree is built into: node refs and leaf item positions are relative to these two vectors, so that subtrees
     * built by different threads can be spliced behind one another and come out numbered as the serial build numbers them */
    struct Out {
        std::vector<Node64> nodes;
        std::vector<uint32_t> order;
        int max_depth = 0;
    };
    Out top;
    std::vector<Node64> &nodes = top.nodes;
    std::vector<uint32_t> &order = top.order;
    int max_depth = 0;
    uint32_t item_base = 0;
    static const int NB = 32;
    float c_trav = 1.0f;  /* SAH: cost of one binary split relative to one item test */
    size_t always_leaf = 2; /* ranges of at most this many items are leaves without asking the SAH */
    int par_levels = 4;     /* the two halves of a split are built by two threads down to this depth (PTR_BUILD_THREADS=1: serial) */
    static const size_t PAR_MIN = 1 << 14;

    Builder(const std::vector<Box> &b, size_t /*nbase*/, size_t ibase) : boxes(b), item_base((uint32_t)ibase) {
        idx.resize(b.size());
        cen.resize(3 * b.size());
        for (size_t i = 0; i < b.size(); i++) {
            idx[i] = (uint32_t)i;
            for (int a = 0; a < 3; a++) cen[3 * i + a] = 0.5f * (b[i].mn[a] + b[i].mx[a]);
        }
        if (const char *e = getenv("PTR_SAH_CT")) c_trav = (float)atof(e);
        if (const char *e = getenv("PTR_SAH_ALWAYS_LEAF")) always_leaf = (size_t)atoi(e);
        if (const char *e = getenv("PTR_BUILD_THREADS")) {
            int t = atoi(e);
            par_levels = 0;
            while ((1 << (par_levels + 1)) <= t) par_levels++;
        }
    }
    static int32_t leaf_ref(uint32_t first, uint32_t n) { return -(int32_t)(first | ((n - 1) << 27)) - 1; }
    static int32_t make_leaf(const std::vector<uint32_t> &idx, size_t lo, size_t hi, Out &out) {
        uint32_t first = (uint32_t)out.order.size(), n = (uint32_t)(hi - lo);
        for (size_t i = lo; i < hi; i++) out.order.push_back(idx[i]);
        return leaf_ref(first, n);
    }
    /* a ref of a subtree that is being appended behind node_off nodes and item_off items */
    static int32_t moved(int32_t ref, int32_t node_off, uint32_t item_off) {
        if (ref >= 0) return ref + node_off;
        const uint32_t packed = (uint32_t)(-(ref + 1));
        return leaf_ref((packed & 0x07FFFFFFu) + item_off, (packed >> 27) + 1);
    }
    static int32_t splice(Out &into, const Out &sub, int32_t sub_root) {
        const int32_t node_off = (int32_t)into.nodes.size();
        const uint32_t item_off = (uint32_t)into.order.size();
        for (Node64 nd : sub.nodes) {
            nd.left = moved(nd.left, node_off, item_off);
            nd.right = moved(nd.right, node_off, item_off);
            into.nodes.push_back(nd);
        }
        into.order.insert(into.order.end(), sub.order.begin(), sub.order.end());
        if (sub.max_depth > into.max_depth) into.max_depth = sub.max_depth;
        return moved(sub_root, node_off, item_off);
    }
    /* the whole tree: refs of inner nodes index `nodes`, leaf refs carry item_base */
    int32_t build(size_t lo, size_t hi, Box *outBox, int depth) {
        int32_t root = build_range(lo, hi, outBox, depth, top, par_levels);
        max_depth = top.max_depth;
        for (Node64 &nd : top.nodes) {
            nd.left = moved(nd.left, 0, item_base);
            nd.right = moved(nd.right, 0, item_base);
        }
        return moved(root, 0, item_base);
    }
    /* fn(chunk, begin, end) over [lo, hi) cut into `chunks` pieces, one thread each (the caller's among them).  What the
     * pieces compute is merged with min / max / integer sums only, so the result does not depend on the cut. */
    template <class F>
    static void for_chunks(size_t lo, size_t hi, int chunks, const F &fn) {
        if (chunks <= 1) { fn(0, lo, hi); return; }
        std::vector<std::thread> th;
        const size_t n = hi - lo;
        int started = 1;
        for (int c = 1; c < chunks; c++) {
            try { th.emplace_back([&, c] { fn(c, lo + n * c / chunks, lo + n * (c + 1) / chunks); }); }
            catch (const std::system_error &) { break; } /* no more threads to be had: the caller takes the rest */
            started++;
        }
        fn(0, lo, lo + n / chunks);
        for (int c = started; c < chunks; c++) fn(c, lo + n * c / chunks, lo + n * (c + 1) / chunks);
        for (std::thread &t : th) t.join();
    }
    static const int MAX_CHUNKS = 16;
    /* returns the child ref for items [lo, hi) and their box */
    int32_t build_range(size_t lo, size_t hi, Box *outBox, int depth, Out &out, int par) {
        size_t n = hi - lo;
        /* the sweeps over a large range near the root, where there are fewer subtrees than threads, are cut into pieces */
        int chunks = (par > 0 && n >= 8 * PAR_MIN) ? (1 << par) : 1;
        if (chunks > MAX_CHUNKS) chunks = MAX_CHUNKS;
        Box bb, cb;
        bb.reset();
        cb.reset();
        {
            Box pbb[MAX_CHUNKS], pcb[MAX_CHUNKS];
            for_chunks(lo, hi, chunks, [&](int c, size_t b0, size_t b1) {
                Box x, y;
                x.reset();
                y.reset();
                for (size_t i = b0; i < b1; i++) {
                    x.grow(boxes[idx[i]]);
                    y.grow(&cen[3 * idx[i]]);
                }
                pbb[c] = x;
                pcb[c] = y;
            });
            for (int c = 0; c < chunks; c++) { bb.grow(pbb[c]); cb.grow(pcb[c]); }
        }
        *outBox = bb;
        if (depth > out.max_depth) out.max_depth = depth;
        if (n <= always_leaf) return make_leaf(idx, lo, hi, out);

        size_t mid = 0;
        bool have_split = false;
        float best_cost = FLT_MAX;
        if (depth <= 28) {
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; a++) {
                float ext = cb.mx[a] - cb.mn[a];
                if (!(ext > 0)) continue;
                Box bin_box[NB];
                uint32_t bin_n[NB];
                for (int b = 0; b < NB; b++) { bin_box[b].reset(); bin_n[b] = 0; }
                float scale = NB / ext;
                if (chunks == 1) {
                    for (size_t i = lo; i < hi; i++) {
                        int b = (int)((cen[3 * idx[i] + a] - cb.mn[a]) * scale);
                        if (b < 0) b = 0;
                        if (b > NB - 1) b = NB - 1;
                        bin_box[b].grow(boxes[idx[i]]);
                        bin_n[b]++;
                    }
                } else {
                    std::vector<Box> pbox((size_t)chunks * NB);
                    std::vector<uint32_t> pn((size_t)chunks * NB, 0u);
                    for (Box &x : pbox) x.reset();
                    for_chunks(lo, hi, chunks, [&](int c, size_t b0, size_t b1) {
                        Box *mybox = &pbox[(size_t)c * NB];
                        uint32_t *myn = &pn[(size_t)c * NB];
                        for (size_t i = b0; i < b1; i++) {
                            int b = (int)((cen[3 * idx[i] + a] - cb.mn[a]) * scale);
                            if (b < 0) b = 0;
                            if (b > NB - 1) b = NB - 1;
                            mybox[b].grow(boxes[idx[i]]);
                            myn[b]++;
                        }
                    });
                    for (int c = 0; c < chunks; c++)
                        for (int b = 0; b < NB; b++) { bin_box[b].grow(pbox[(size_t)c * NB + b]); bin_n[b] += pn[(size_t)c * NB + b]; }
                }
                float right_area[NB];
                uint32_t right_n[NB];
                Box acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int b = NB - 1; b > 0; b--) {
                    acc.grow(bin_box[b]);
                    cnt += bin_n[b];
                    right_area[b] = acc.half_area();
                    right_n[b] = cnt;
                }
                acc.reset();
                cnt = 0;
                for (int b = 0; b < NB - 1; b++) {
                    acc.grow(bin_box[b]);
                    cnt += bin_n[b];
                    if (cnt == 0 || right_n[b + 1] == 0) continue;
                    float cost = acc.half_area() * cnt + right_area[b + 1] * right_n[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                float parent = bb.half_area();
                float split_cost = c_trav + (parent > 0 ? best_cost / parent : (float)n);
                if (n <= BVH_MAX_LEAF && (float)n <= split_cost) return make_leaf(idx, lo, hi, out);
                float ext = cb.mx[best_axis] - cb.mn[best_axis];
                float scale = NB / ext;
                float cmin = cb.mn[best_axis];
                auto it = std::partition(idx.begin() + lo, idx.begin() + hi, [&](uint32_t id) {
                    int b = (int)((cen[3 * id + best_axis] - cmin) * scale);
                    if (b < 0) b = 0;
                    if (b > NB - 1) b = NB - 1;
                    return b <= best_bin;
                });
                mid = (size_t)(it - idx.begin());
                have_split = mid > lo && mid < hi;
            }
        }
        if (!have_split) {
            if (n <= BVH_MAX_LEAF) return make_leaf(idx, lo, hi, out);
            /* coincident centroids or depth guard: split the index range in half on the widest axis */
            int a = 0;
            float e0 = cb.mx[0] - cb.mn[0], e1 = cb.mx[1] - cb.mn[1], e2 = cb.mx[2] - cb.mn[2];
            if (e1 > e0 && e1 >= e2) a = 1;
            else if (e2 > e0 && e2 > e1) a = 2;
            mid = lo + n / 2;
            std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi,
                             [&](uint32_t x, uint32_t y) { return cen[3 * x + a] < cen[3 * y + a]; });
        }
        int32_t me = (int32_t)out.nodes.size();
        out.nodes.emplace_back();
        Box lb, rb;
        int32_t l, r;
        if (par > 0 && n >= PAR_MIN) {
            /* the halves are disjoint ranges of idx: the left one in another thread, each into its own vectors, appended in the
             * serial build's order (this node, the left subtree, the right subtree) */
            Out lo_out, hi_out;
            int32_t l_rel = 0;
            std::thread th;
            bool spawned = true;
            try { th = std::thread([&] { l_rel = build_range(lo, mid, &lb, depth + 1, lo_out, par - 1); }); }
            catch (const std::system_error &) { spawned = false; } /* no thread to be had: this one builds both halves */
            const int32_t r_rel = build_range(mid, hi, &rb, depth + 1, hi_out, par - 1);
            if (spawned) th.join();
            else l_rel = build_range(lo, mid, &lb, depth + 1, lo_out, 0);
            l = splice(out, lo_out, l_rel);
            r = splice(out, hi_out, r_rel);
        } else {
            l = build_range(lo, mid, &lb, depth + 1, out, 0);
            r = build_range(mid, hi, &rb, depth + 1, out, 0);
        }
        Node64 &nd = out.nodes[me];
        for (int i = 0; i < 3; i++) { nd.lmin[i] = lb.mn[i]; nd.lmax[i] = lb.mx[i]; nd.rmin[i] = rb.mn[i]; nd.rmax[i] = rb.mx[i]; }
        nd.left = l;
        nd.right = r;
        return me;
    }
};

/* Binary SAH tree -> 4-wide nodes: starting from a node's two children, the inner child with the
 * largest surface area is replaced by its own two children until four slots are taken (or only
 * leaves remain).  Leaves and their item ranges are kept as they are. */
struct Collapser {
    const std::vector<Node64> &bin; /* refs >= 0 index this vector */
    std::vector<Node128> &out;      /* refs >= 0 written here index this vector (global) */
    int max_depth = 0;
    Collapser(const std::vector<Node64> &b, std::vector<Node128> &o) : bin(b), out(o) {}
    struct Slot { Box box; int32_t ref; };
    static Box box_of(const float *mn, const float *mx) {
        Box b;
        for (int a = 0; a < 3; a++) { b.mn[a] = mn[a]; b.mx[a] = mx[a]; }
        return b;
    }
    int32_t emit(int32_t ref, int depth) {
        if (ref < 0) return ref;
        if (depth > max_depth) max_depth = depth;
        Slot sl[4];
        int n = 2;
        {
            const Node64 &nd = bin[ref];
            sl[0] = Slot{box_of(nd.lmin, nd.lmax), nd.left};
            sl[1] = Slot{box_of(nd.rmin, nd.rmax), nd.right};
        }
        while (n < 4) {
            int pick = -1;
            float area = -1;
            for (int i = 0; i < n; i++)
                if (sl[i].ref >= 0 && sl[i].box.half_area() > area) { area = sl[i].box.half_area(); pick = i; }
            if (pick < 0) break;
            const Node64 &ch = bin[sl[pick].ref];
            sl[pick] = Slot{box_of(ch.lmin, ch.lmax), ch.left};
            sl[n++] = Slot{box_of(ch.rmin, ch.rmax), ch.right};
        }
        const size_t me = out.size();
        out.emplace_back();
        int32_t refs[4];
        for (int i = 0; i < n; i++) refs[i] = emit(sl[i].ref, depth + 1);
        Node128 &o = out[me];
        memset(&o, 0, sizeof(o));
        for (int i = 0; i < 4; i++) {
            if (i < n) {
                for (int a = 0; a < 3; a++) { o.c[i].mn[a] = sl[i].box.mn[a]; o.c[i].mx[a] = sl[i].box.mx[a]; }
                o.c[i].ref = refs[i];
            } else o.c[i].ref = BVH_NO_CHILD;
        }
        return (int32_t)me;
    }
};

/* ---- the reference's top-level visiting order ------------------------------------------ */
inline float host_rand_float() { return rand() / 2147483648.f; } /* util.h:64-66: rand() / (RAND_MAX + 1.f) */
struct TopHandle {
    float pmin[3], pmax[3];
    int32_t prim;
};
/* util.h:10-39 with CompareX/Y/Z (a.bound.pMin[axis] >= b.bound.pMin[axis]) */
void sort_handles(TopHandle *arr, int elements, int axis) {
    const int MAX_LEVELS = 1000;
    int beg[MAX_LEVELS], end[MAX_LEVELS], i = 0;
    beg[0] = 0;
    end[0] = elements;
    while (i >= 0) {
        int L = beg[i], R = end[i] - 1;
        if (L < R) {
            TopHandle piv = arr[L];
            if (i == MAX_LEVELS - 1) return;
            while (L < R) {
                while (arr[R].pmin[axis] >= piv.pmin[axis] && L < R) R--;
                if (L < R) arr[L++] = arr[R];
                while (piv.pmin[axis] >= arr[L].pmin[axis] && L < R) L++;
                if (L < R) arr[R--] = arr[L];
            }
            arr[L] = piv;
            beg[i + 1] = L + 1;
            end[i + 1] = end[i];
            end[i++] = L;
        } else i--;
    }
}
/* _CreateBVH (primitive.cpp:537-582) over the primitives: median split, sort axis from rand(),
 * sort skipped when the axis repeats; rand_float() can round to 1.0f, axis 3 sorts nothing */
int build_top(std::vector<TopNode> &nodes, std::vector<int32_t> &order, TopHandle *h, int n, int depth, int max_depth, int last_axis) {
    int me = (int)nodes.size();
    nodes.emplace_back();
    int axis = int(3 * host_rand_float());
    if (axis != last_axis) {
        last_axis = axis;
        if (axis >= 0 && axis <= 2) sort_handles(h, n, axis);
    }
    TopNode nd;
    memset(&nd, 0, sizeof(nd));
    if (n == 1 || depth >= max_depth) {
        nd.left = nd.right = -1;
        nd.first = (int32_t)order.size();
        nd.n = n;
        for (int a = 0; a < 3; a++) { nd.bmin[a] = h[0].pmin[a]; nd.bmax[a] = h[0].pmax[a]; }
        for (int i = 0; i < n; i++) {
            order.push_back(h[i].prim);
            for (int a = 0; a < 3; a++) {
                if (h[i].pmin[a] < nd.bmin[a]) nd.bmin[a] = h[i].pmin[a];
                if (h[i].pmax[a] > nd.bmax[a]) nd.bmax[a] = h[i].pmax[a];
            }
        }
        nodes[me] = nd;
        return me;
    }
    int l = build_top(nodes, order, h, n / 2, depth + 1, max_depth, last_axis);
    int r = build_top(nodes, order, h + n / 2, n - n / 2, depth + 1, max_depth, last_axis);
    nd.left = l;
    nd.right = r;
    for (int a = 0; a < 3; a++) {
        nd.bmin[a] = nodes[l].bmin[a] < nodes[r].bmin[a] ? nodes[l].bmin[a] : nodes[r].bmin[a];
        nd.bmax[a] = nodes[l].bmax[a] > nodes[r].bmax[a] ? nodes[l].bmax[a] : nodes[r].bmax[a];
    }
    nodes[me] = nd;
    return me;
}
/* ---- a second tree over the same primitives, built for speed (round 3) ------------------------------------------
 * The reference's tree splits at the median along axes drawn from rand(): in a room of a hundred primitives a ray steps through
 * ~25 of its nodes.  Most rays do not need its order: an any-hit ray's answer cannot depend on the order at all (its tMax never
 * changes before the walk ends), and a closest-hit ray's answer is the candidate of smallest t unless the reference would have
 * culled it or two candidates lie within rounding of each other - which the tracer detects, and then walks the reference's tree
 * (dev_trace.h "fast top level").  This tree is a plain full-sweep SAH binary tree, one primitive per leaf, over boxes that hold
 * every hit point a primitive can report: the shape's bounds, for a rectangle also the slab its test really uses
 * (rectangle.cpp:8 vs :37-40: thicker than its bounds), grown by a margin far above rounding. */
struct FastHandle {
    float mn[3], mx[3], c[3];
    int32_t prim;
};
inline float box_area(const float *mn, const float *mx) {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2.f * (dx * dy + dy * dz + dz * dx);
}
int build_fast_top(std::vector<TopNode> &nodes, std::vector<int32_t> &order, FastHandle *h, int n) {
    const int me = (int)nodes.size();
    nodes.emplace_back();
    TopNode nd;
    memset(&nd, 0, sizeof(nd));
    for (int a = 0; a < 3; a++) { nd.bmin[a] = h[0].mn[a]; nd.bmax[a] = h[0].mx[a]; }
    for (int i = 1; i < n; i++)
        for (int a = 0; a < 3; a++) {
            if (h[i].mn[a] < nd.bmin[a]) nd.bmin[a] = h[i].mn[a];
            if (h[i].mx[a] > nd.bmax[a]) nd.bmax[a] = h[i].mx[a];
        }
    if (n == 1) {
        nd.left = nd.right = -1;
        nd.first = (int32_t)order.size();
        nd.n = 1;
        order.push_back(h[0].prim);
        nodes[me] = nd;
        return me;
    }
    /* full sweep along each axis over the handles sorted by box centre: cost = area(L) * |L| + area(R) * |R| */
    int best_axis = 0, best_k = n / 2;
    float best_cost = FLT_MAX;
    std::vector<float> right_area(n);
    for (int axis = 0; axis < 3; axis++) {
        std::stable_sort(h, h + n, [axis](const FastHandle &a, const FastHandle &b) { return a.c[axis] < b.c[axis]; });
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int i = n - 1; i >= 1; i--) {
            for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], h[i].mn[a]); mx[a] = std::max(mx[a], h[i].mx[a]); }
            right_area[i] = box_area(mn, mx);
        }
        for (int a = 0; a < 3; a++) { mn[a] = FLT_MAX; mx[a] = -FLT_MAX; }
        for (int k = 1; k < n; k++) { /* left = [0, k), right = [k, n) */
            for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], h[k - 1].mn[a]); mx[a] = std::max(mx[a], h[k - 1].mx[a]); }
            const float cost = box_area(mn, mx) * k + right_area[k] * (n - k);
            if (cost < best_cost) { best_cost = cost; best_axis = axis; best_k = k; }
        }
    }
    std::stable_sort(h, h + n, [best_axis](const FastHandle &a, const FastHandle &b) { return a.c[best_axis] < b.c[best_axis]; });
    const int l = build_fast_top(nodes, order, h, best_k);
    const int r = build_fast_top(nodes, order, h + best_k, n - best_k);
    nd.left = l;
    nd.right = r;
    nodes[me] = nd;
    return me;
}
/* the box of primitive `pr` in the fast tree; false: not finite (the scene then keeps the reference's tree only) */
bool fast_top_box(const DPrim &pr, FastHandle *out) {
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) { mn[a] = pr.bmin[a]; mx[a] = pr.bmax[a]; }
    if (pr.type == SH_RECT) { /* the eight corners of the slab of Rectangle::Intersect, in world space */
        const float hx = pr.p[0] / 2, hy = pr.p[1] / 2, zb = 0.0001f * std::max(hx, hy);
        for (int k = 0; k < 8; k++) {
            const float x = (k & 1) ? hx : -hx, y = (k & 2) ? hy : -hy, z = (k & 4) ? zb : -zb;
            for (int a = 0; a < 3; a++) {
                const float w = pr.o2w[4 * a] * x + pr.o2w[4 * a + 1] * y + pr.o2w[4 * a + 2] * z + pr.o2w[4 * a + 3];
                mn[a] = std::min(mn[a], w);
                mx[a] = std::max(mx[a], w);
            }
        }
    }
    float ext = 0, mag = 0;
    for (int a = 0; a < 3; a++) {
        if (!std::isfinite(mn[a]) || !std::isfinite(mx[a]) || mn[a] > mx[a]) return false;
        ext = std::max(ext, mx[a] - mn[a]);
        mag = std::max(mag, std::max(std::fabs(mn[a]), std::fabs(mx[a])));
    }
    const float margin = 1e-4f * ext + 1e-5f * mag + 1e-6f;
    if (!std::isfinite(margin)) return false;
    for (int a = 0; a < 3; a++) {
        out->mn[a] = mn[a] - margin;
        out->mx[a] = mx[a] + margin;
        out->c[a] = 0.5f * mn[a] + 0.5f * mx[a];
        if (!std::isfinite(out->mn[a]) || !std::isfinite(out->mx[a])) return false;
    }
    return true;
}
/* Numbers the nodes of `tree` breadth first - inner nodes first, then the leaves, so that the LDS copy of the first TOP_LDS_NODES
 * holds only nodes the persistent tracer reads - and gives every inner node copies of its children's boxes and, for a child that
 * is a one-primitive leaf, that primitive (TopNode::pad: the tracer then goes from the parent straight to the primitive).  Child
 * indices are offset by `base`, the place of the tree's root in the scene's node array; `order` = the tree's leaf handles. */
void finish_top_tree(std::vector<TopNode> &tree, const std::vector<int32_t> &order, int32_t base) {
    if (tree.empty()) return;
    const size_t nt = tree.size();
    std::vector<int32_t> bfs{0}, new_of(nt, -1);
    for (size_t head = 0; head < bfs.size(); head++) {
        const TopNode &nd = tree[bfs[head]];
        if (nd.left >= 0) { bfs.push_back(nd.left); bfs.push_back(nd.right); }
    }
    int32_t next = 0;
    if (tree[0].left < 0) new_of[0] = next++; /* a tree of one leaf: the root stays node 0 */
    for (size_t k = 0; k < bfs.size(); k++)
        if (tree[bfs[k]].left >= 0) new_of[bfs[k]] = next++;
    for (size_t k = 0; k < bfs.size(); k++)
        if (new_of[bfs[k]] < 0) new_of[bfs[k]] = next++;
    std::vector<TopNode> moved(nt);
    for (size_t i = 0; i < nt; i++) {
        TopNode nd = tree[i];
        nd.pad[0] = nd.pad[1] = -1;
        if (nd.left >= 0) {
            const TopNode &l = tree[nd.left], &r = tree[nd.right];
            for (int a = 0; a < 3; a++) { nd.lmin[a] = l.bmin[a]; nd.lmax[a] = l.bmax[a]; nd.rmin[a] = r.bmin[a]; nd.rmax[a] = r.bmax[a]; }
            if (l.left < 0 && l.n == 1) nd.pad[0] = order[l.first];
            if (r.left < 0 && r.n == 1) nd.pad[1] = order[r.first];
            nd.left = base + new_of[nd.left];
            nd.right = base + new_of[nd.right];
        }
        moved[new_of[i]] = nd;
    }
    tree.swap(moved);
}

/* the rand() draws of one _CreateBVH over n handles: one per node */
void burn_bvh_rand(int n, int depth, int max_depth) {
    if (n <= 0) return;
    (void)rand();
    if (n == 1 || depth >= max_depth) return;
    burn_bvh_rand(n / 2, depth + 1, max_depth);
    burn_bvh_rand(n - n / 2, depth + 1, max_depth);
}

int fail(std::string *err, int code, const std::string &msg) {
    if (err) *err = msg;
    return code;
}

} // namespace

static int g_bvh_mode = -1;
int ptr_bvh_build_mode_get() {
    if (g_bvh_mode < 0) {
        const char *e = getenv("PTR_BVH_BUILD");
        g_bvh_mode = (e && !strcmp(e, "device")) ? PTR_BVH_DEVICE_LBVH : PTR_BVH_HOST_SAH;
    }
    return g_bvh_mode;
}
void ptr_bvh_build_mode_set(int mode) { g_bvh_mode = mode == PTR_BVH_DEVICE_LBVH ? PTR_BVH_DEVICE_LBVH : PTR_BVH_HOST_SAH; }

/* Bounds3::BoundingSphere (geometry.h) of the world bound: radius = Distance(center, pMax) if the centre is inside, else 0 */
template <class WorldBox> static float world_radius(const WorldBox &world) {
    float c[3];
    bool inside = true;
    for (int a = 0; a < 3; a++) {
        c[a] = (world.mn[a] + world.mx[a]) * (1.0f / 2);
        inside = inside && c[a] >= world.mn[a] && c[a] <= world.mx[a];
    }
    const float dx = c[0] - world.mx[0], dy = c[1] - world.mx[1], dz = c[2] - world.mx[2];
    return inside ? sqrtf(dx * dx + dy * dy + dz * dz) : 0;
}
int ptr_flatten(const PtrFlatScene *fs, HostScene *hs, std::string *err) {
    if (!fs || !hs) return fail(err, PTR_ERR_INVALID, "null scene");
    if (fs->n_primitives < 0 || fs->n_meshes < 0 || fs->n_clouds < 0 || fs->n_images < 0)
        return fail(err, PTR_ERR_INVALID, "negative count in PtrFlatScene");
    if ((fs->n_primitives > 0 && !fs->primitives) || (fs->n_meshes > 0 && !fs->meshes) || (fs->n_clouds > 0 && !fs->clouds) || (fs->n_images > 0 && !fs->images))
        return fail(err, PTR_ERR_INVALID, "null array with a non-zero count in PtrFlatScene");
    *hs = HostScene();

    for (int i = 0; i < fs->n_images; i++) {
        const PtrImage &pi = fs->images[i];
        if (pi.width <= 0 || pi.height <= 0 || !pi.rgb) return fail(err, PTR_ERR_INVALID, "bad image");
        HostImage im;
        im.w = pi.width;
        im.h = pi.height;
        im.rgb.assign(pi.rgb, pi.rgb + 3 * (size_t)pi.width * pi.height);
        hs->images.push_back(std::move(im));
    }
    /* meshes: vertices to world space (mesh.cpp:55-65), triangles pre-gathered */
    for (int i = 0; i < fs->n_meshes; i++) {
        const PtrMesh &pm = fs->meshes[i];
        if (pm.n_triangles < 0 || pm.n_vertices < 0 || (pm.n_triangles > 0 && (!pm.p || !pm.indices)))
            return fail(err, PTR_ERR_INVALID, "bad mesh arrays");
        if (!is_affine(pm.to_world.m) || !is_affine(pm.to_world.mInv)) return fail(err, PTR_ERR_LIMIT, "projective mesh transform");
        HostMesh hm;
        hm.tri_base = (int64_t)(hs->tris.size() / 9);
        hm.n_tris = pm.n_triangles;
        hm.n_uvs = pm.uv ? pm.n_uvs : 0;
        std::vector<V3h> wp(pm.n_vertices);
        for (int v = 0; v < pm.n_vertices; v++) {
            V3h p = {pm.p[3 * v], pm.p[3 * v + 1], pm.p[3 * v + 2]};
            wp[v] = xf_point16(pm.to_world.m, p);
        }
        hs->tris.reserve(hs->tris.size() + 9 * (size_t)pm.n_triangles);
        for (int t = 0; t < pm.n_triangles; t++) {
            for (int c = 0; c < 3; c++) {
                int vi = pm.indices[9 * (size_t)t + 3 * c];
                if (vi < 0 || vi >= pm.n_vertices) return fail(err, PTR_ERR_INVALID, "mesh vertex index out of range");
                hs->tris.push_back(wp[vi].x);
                hs->tris.push_back(wp[vi].y);
                hs->tris.push_back(wp[vi].z);
                hs->uvidx.push_back(pm.indices[9 * (size_t)t + 3 * c + 2]);
            }
        }
        if (pm.uv) {
            hm.uv.assign(pm.uv, pm.uv + 2 * (size_t)pm.n_uvs);
            hm.uv.push_back(0.f);
            hm.uv.push_back(0.f);
        }
        hs->meshes.push_back(std::move(hm));
    }
    for (int i = 0; i < fs->n_clouds; i++) {
        const PtrCloud &pc = fs->clouds[i];
        if (pc.n_particles < 0 || (pc.n_particles > 0 && !pc.positions)) return fail(err, PTR_ERR_INVALID, "bad particle cloud");
        HostCloud hc;
        hc.n = pc.n_particles;
        hc.radius = pc.radius;
        hc.pos.assign(pc.positions, pc.positions + 3 * (size_t)pc.n_particles);
        hs->clouds.push_back(std::move(hc));
    }

    /* one SAH BVH per mesh and per particle cloud, all in the same node / item arrays */
    std::vector<int32_t> mesh_root(hs->meshes.size(), BVH_NO_CHILD), cloud_root(hs->clouds.size(), BVH_NO_CHILD);
    std::vector<Box> mesh_box(hs->meshes.size()), cloud_box(hs->clouds.size());
    hs->bvh_builder = ptr_bvh_build_mode_get();
    auto build_geom_inner = [&](const std::vector<Box> &boxes, const std::vector<Item48> &raw, int32_t *root, Box *bound) -> int {
        if (boxes.empty()) return PTR_OK;
        if (hs->items.size() + boxes.size() >= (1u << 27)) return fail(err, PTR_ERR_LIMIT, "more than 2^27 BVH items");
        if (hs->bvh_builder == PTR_BVH_DEVICE_LBVH) { /* linear BVH on the current device, same tables (rt_build.hip) */
            static_assert(sizeof(Box) == 6 * sizeof(float), "Box is six floats");
            int depth = 0;
            double ms = 0;
            std::string gerr;
            int rc = ptr_gpu_build_geom(reinterpret_cast<const float *>(boxes.data()), raw.data(), boxes.size(), &hs->nodes, &hs->items, root, bound->mn, &depth, &ms,
                                        &gerr);
            if (rc) return fail(err, rc, gerr);
            hs->bvh_build_ms += ms;
            if (3 * (depth + 1) + 24 > BVH_STACK) return fail(err, PTR_ERR_LIMIT, "BVH deeper than the traversal stack");
            if (depth > hs->bvh_depth) hs->bvh_depth = depth;
            return PTR_OK;
        }
        const auto t_begin = std::chrono::steady_clock::now();
        Builder b(boxes, 0, hs->items.size());
        b.nodes.reserve(boxes.size());
        b.order.reserve(boxes.size());
        const int32_t root2 = b.build(0, boxes.size(), bound, 0);
        const auto t_mid = std::chrono::steady_clock::now();
        Collapser c(b.nodes, hs->nodes);
        *root = c.emit(root2, 0);
        if (getenv("PTR_BUILD_VERBOSE")) fprintf(stderr, "sah build %zu items: binary tree %.1f ms, collapse %.1f ms\n", boxes.size(),
            std::chrono::duration<double, std::milli>(t_mid - t_begin).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_mid).count());
        /* per lane: up to three pushed siblings per level of this tree, on top of the top-level walk's entries
         * (the persistent tracer keeps both in one stack; the reference's top tree is at most 20 deep) */
        if (3 * (c.max_depth + 1) + 24 > BVH_STACK) return fail(err, PTR_ERR_LIMIT, "BVH deeper than the traversal stack");
        if (c.max_depth > hs->bvh_depth) hs->bvh_depth = c.max_depth;
        for (size_t i = 0; i < b.order.size(); i++) hs->items.push_back(raw[b.order[i]]);
        hs->bvh_build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        return PTR_OK;
    };
    auto build_geom = [&](const std::vector<Box> &boxes, const std::vector<Item48> &raw, int32_t *root, Box *bound) -> int {
        int rc = build_geom_inner(boxes, raw, root, bound);
        if (rc || boxes.size() <= hs->sah_of_items || *root < 0) return rc;
        /* expected work per ray through the root box (uniform random lines: P(child | root) = area ratio) */
        hs->sah_of_items = boxes.size();
        hs->sah_boxes = hs->sah_items = hs->sah_leaves = 0;
        const float root_area = bound->half_area();
        if (!(root_area > 0)) return rc;
        std::vector<std::pair<int32_t, double>> todo{{*root, 1.0}};
        while (!todo.empty()) {
            const std::pair<int32_t, double> cur = todo.back();
            todo.pop_back();
            const Node128 &nd = hs->nodes[cur.first];
            for (int i = 0; i < 4; i++) {
                if (nd.c[i].ref == BVH_NO_CHILD) continue;
                hs->sah_boxes += cur.second;
                Box cb;
                for (int a = 0; a < 3; a++) { cb.mn[a] = nd.c[i].mn[a]; cb.mx[a] = nd.c[i].mx[a]; }
                const double p = cb.half_area() / root_area;
                if (nd.c[i].ref >= 0) todo.push_back({nd.c[i].ref, p});
                else {
                    hs->sah_leaves += p;
                    hs->sah_items += p * ((((uint32_t)(-(nd.c[i].ref + 1))) >> 27) + 1);
                }
            }
        }
        return rc;
    };
    for (size_t mi = 0; mi < hs->meshes.size(); mi++) {
        const HostMesh &hm = hs->meshes[mi];
        std::vector<Box> boxes(hm.n_tris);
        std::vector<Item48> raw(hm.n_tris);
        for (int t = 0; t < hm.n_tris; t++) { /* MeshGetBounds, mesh.cpp:609-626 */
            const float *v = &hs->tris[9 * (size_t)(hm.tri_base + t)];
            Box tb;
            tb.reset();
            tb.grow(v);
            tb.grow(v + 3);
            tb.grow(v + 6);
            Item48 it;
            it.q[0] = v[0]; it.q[1] = v[1]; it.q[2] = v[2]; it.q[3] = 0;
            it.q[4] = v[3]; it.q[5] = v[4]; it.q[6] = v[5]; it.q[7] = as_float(t);
            it.q[8] = v[6]; it.q[9] = v[7]; it.q[10] = v[8]; it.q[11] = as_float(ITEM_TRI);
            boxes[t] = tb;
            raw[t] = it;
        }
        int rc = build_geom(boxes, raw, &mesh_root[mi], &mesh_box[mi]);
        if (rc) return rc;
    }
    for (size_t ci = 0; ci < hs->clouds.size(); ci++) {
        const HostCloud &c = hs->clouds[ci];
        std::vector<Box> boxes(c.n);
        std::vector<Item48> raw(c.n);
        for (int k = 0; k < c.n; k++) { /* ParticleCloudGetBounds, particlecloud.cpp:284-293 */
            Box tb;
            for (int a = 0; a < 3; a++) {
                tb.mn[a] = -c.radius + c.pos[3 * (size_t)k + a];
                tb.mx[a] = c.radius + c.pos[3 * (size_t)k + a];
            }
            Item48 it;
            memset(&it, 0, sizeof(it));
            it.q[0] = c.pos[3 * (size_t)k]; it.q[1] = c.pos[3 * (size_t)k + 1]; it.q[2] = c.pos[3 * (size_t)k + 2];
            it.q[4] = c.radius;
            it.q[7] = as_float(k);
            it.q[11] = as_float(ITEM_PARTICLE);
            boxes[k] = tb;
            raw[k] = it;
        }
        int rc = build_geom(boxes, raw, &cloud_root[ci], &cloud_box[ci]);
        if (rc) return rc;
    }

    /* The upper levels of all geometry trees first in nodes[] (breadth first, the trees taking turns), the rest in the
     * order the builder left them: the tracer can keep nodes[0 .. NODE_PREFIX) in LDS (dev_trace.h, PT_LDS_TOP), and the
     * nodes every ray starts with share a few cache lines either way. */
    {
        const size_t nn = hs->nodes.size();
        std::vector<int32_t> order; /* new index -> old index, the prefix */
        std::vector<int32_t> queue;
        for (int32_t r : mesh_root) if (r >= 0) queue.push_back(r);
        for (int32_t r : cloud_root) if (r >= 0) queue.push_back(r);
        for (size_t head = 0; head < queue.size() && order.size() < (size_t)BVH_NODE_PREFIX; head++) {
            const int32_t cur = queue[head];
            order.push_back(cur);
            for (int i = 0; i < 4; i++) {
                const int32_t ref = hs->nodes[cur].c[i].ref;
                if (ref >= 0 && ref != BVH_NO_CHILD) queue.push_back(ref);
            }
        }
        std::vector<int32_t> new_of(nn, -1);
        for (size_t k = 0; k < order.size(); k++) new_of[order[k]] = (int32_t)k;
        int32_t next = (int32_t)order.size();
        for (size_t i = 0; i < nn; i++) if (new_of[i] < 0) new_of[i] = next++;
        std::vector<Node128> moved(nn);
        for (size_t i = 0; i < nn; i++) {
            Node128 nd = hs->nodes[i];
            for (int k = 0; k < 4; k++)
                if (nd.c[k].ref >= 0) nd.c[k].ref = new_of[nd.c[k].ref];
            moved[new_of[i]] = nd;
        }
        hs->nodes.swap(moved);
        for (int32_t &r : mesh_root) if (r >= 0) r = new_of[r];
        for (int32_t &r : cloud_root) if (r >= 0) r = new_of[r];
    }

    if (PT_QNODES) quantise_nodes(hs->nodes, &hs->qnodes); /* only a build whose tracer reads them */

    if (fs->camera_medium.is_valid) {
        hs->view_medium = (int)hs->media.size();
        hs->media.push_back(conv_medium(fs->camera_medium));
        hs->has_media = true;
    }

    Box world;
    world.reset();

    for (int i = 0; i < fs->n_primitives; i++) {
        const PtrPrimitive &pp = fs->primitives[i];
        DPrim pr;
        memset(&pr, 0, sizeof(pr));
        pr.type = pp.shape_type;
        pr.geom = pp.geom;
        pr.material = -1;
        pr.light = -1;
        pr.med_inside = pr.med_outside = -1;
        const PtrTransform *tw = &pp.to_world;
        PtrTransform ident;
        bool reverse = pp.reverse_orientation != 0;
        float lmin[3] = {0, 0, 0}, lmax[3] = {0, 0, 0};
        bool ok = true;
        switch (pp.shape_type) {
        case PTR_SHAPE_SPHERE: /* shape.h:74-78 */
            pr.p[0] = pp.params[0];
            pr.p[1] = -pr.p[0];
            pr.p[2] = pr.p[0];
            pr.p[3] = acosf(-1.f);
            pr.p[4] = acosf(1.f);
            pr.p[5] = radians_f(360);
            lmin[0] = lmin[1] = -pr.p[0]; lmin[2] = pr.p[1];
            lmax[0] = lmax[1] = pr.p[0]; lmax[2] = pr.p[2];
            pr.area = pr.p[5] * pr.p[0] * (pr.p[2] - pr.p[1]); /* sphere.cpp:21-23 */
            break;
        case PTR_SHAPE_SPHERE_PROCEDURAL:
            pr.p[0] = pp.params[0];
            reverse = false;
            lmin[0] = lmin[1] = lmin[2] = -pr.p[0];
            lmax[0] = lmax[1] = lmax[2] = pr.p[0];
            pr.area = 1; /* procedural.h:59 */
            break;
        case PTR_SHAPE_RECTANGLE: { /* shape.h:119-122, rectangle.cpp:37-44 */
            pr.p[0] = pp.params[0] < 0 ? 0 : pp.params[0];
            pr.p[1] = pp.params[1] < 0 ? 0 : pp.params[1];
            float hx = pr.p[0] / 2, hy = pr.p[1] / 2;
            lmin[0] = -hx; lmin[1] = -hy; lmin[2] = (float)-0.0001;
            lmax[0] = hx; lmax[1] = hy; lmax[2] = (float)0.0001;
            pr.area = pr.p[0] * pr.p[1];
        } break;
        case PTR_SHAPE_DISK: /* shape.h:96-99, disk.cpp:3-6,46-48 */
            pr.p[0] = pp.params[0];
            pr.p[1] = pp.params[1];
            pr.p[2] = pp.params[2];
            pr.p[3] = radians_f(clampf(pp.params[3], 0, 360));
            lmin[0] = lmin[1] = -pr.p[1]; lmin[2] = pr.p[0];
            lmax[0] = lmax[1] = pr.p[1]; lmax[2] = pr.p[0];
            pr.area = (float)(pr.p[3] * 0.5 * (pr.p[1] * pr.p[1] - pr.p[2] * pr.p[2]));
            break;
        case PTR_SHAPE_BOX_PROCEDURAL:
            for (int k = 0; k < 3; k++) {
                pr.p[k] = pp.params[k];
                float half = (float)(pr.p[k] * 0.5);
                lmin[k] = -half;
                lmax[k] = half;
            }
            reverse = false;
            pr.area = 1;
            break;
        case PTR_SHAPE_COMPONENT_PROCEDURAL:
            for (int k = 0; k < 6; k++) pr.p[k] = pp.params[k];
            for (int k = 0; k < 3; k++) { lmin[k] = pp.params[k]; lmax[k] = pp.params[3 + k]; }
            pr.toy_id = pp.geom;
            if (pp.geom >= 0 && pp.geom <= 3) { /* origami: ProceduralComponent::Map's half extents, procedural.cpp:381-382 */
                float half[3];
                for (int k = 0; k < 3; k++) half[k] = fabsf(pp.params[3 + k] - pp.params[k]) * 0.5f;
                pr.root = (int)hs->toy_tris.size();
                pr.geom = ptr_build_toy(pp.geom, half, &hs->toy_tris);
            }
            reverse = false;
            pr.area = 1;
            break;
        case PTR_SHAPE_MESH:
            if (pp.geom < 0 || pp.geom >= fs->n_meshes) return fail(err, PTR_ERR_INVALID, "mesh index out of range");
            tw = &fs->meshes[pp.geom].to_world;
            reverse = false;
            break;
        case PTR_SHAPE_PARTICLECLOUD: {
            if (pp.geom < 0 || pp.geom >= fs->n_clouds) return fail(err, PTR_ERR_INVALID, "cloud index out of range");
            memset(&ident, 0, sizeof(ident));
            ident.m[0] = ident.m[5] = ident.m[10] = ident.m[15] = 1;
            ident.mInv[0] = ident.mInv[5] = ident.mInv[10] = ident.mInv[15] = 1;
            tw = &ident;
            reverse = false;
            const HostCloud &c = hs->clouds[pp.geom];
            pr.p[0] = c.radius;
            float area = 0; /* particlecloud.cpp:16-29 */
            for (int k = 0; k < c.n; k++) area += (float)(4 * kPiD * c.radius * c.radius);
            pr.area = area;
        } break;
        default: ok = false; break; /* "Skipping unknown shape type", scene.cpp:911-915 */
        }
        if (!ok) continue;
        if (!is_affine(tw->m) || !is_affine(tw->mInv)) return fail(err, PTR_ERR_LIMIT, "projective primitive transform");
        copy34(pr.o2w, tw->m);
        copy34(pr.w2o, tw->mInv);
        bool swaps = swaps_handedness(tw->m);
        if (reverse) pr.flags |= PF_REVERSE;
        if (reverse ^ swaps) pr.flags |= PF_FLIP_N;

        Box pb;
        pb.reset();
        if (pp.shape_type != PTR_SHAPE_COMPONENT_PROCEDURAL) pr.root = BVH_NO_CHILD; /* toys: first DToyTri record, set above */
        if (pp.shape_type == PTR_SHAPE_MESH) {
            if (hs->meshes[pp.geom].n_tris == 0) continue; /* an empty mesh has no bounds; the reference would crash */
            pb = mesh_box[pp.geom];
            pr.root = mesh_root[pp.geom];
        } else if (pp.shape_type == PTR_SHAPE_PARTICLECLOUD) {
            if (hs->clouds[pp.geom].n == 0) continue;
            pb = cloud_box[pp.geom];
            pr.root = cloud_root[pp.geom];
        } else {
            pb = xf_bounds16(tw->m, lmin, lmax);
        }
        for (int a = 0; a < 3; a++) { pr.bmin[a] = pb.mn[a]; pr.bmax[a] = pb.mx[a]; }
        world.grow(pb);

        /* material / emission / interior medium, scene.cpp:917-944 */
        int medium = -1;
        if (pp.mat.is_emissive && !pp.no_mat) {
            pr.flags |= PF_EMISSIVE;
            for (int k = 0; k < 3; k++) pr.L[k] = pp.mat.textures[0].spec[k];
        } else {
            if (!pp.no_mat && pp.mat.type >= PTR_MAT_MATTE && pp.mat.type <= PTR_MAT_DISNEY) {
                DMat m;
                memset(&m, 0, sizeof(m));
                m.type = pp.mat.type;
                {
                    static const int lobes_of[] = {0, /* matte */ 1, /* mirror */ 1, /* glass */ 2, /* metal */ 1, /* translucent */ 4, /* plastic */ 2,
                                                   /* uber */ 5, /* subsurface */ 2, /* kdsubsurface */ 2, /* playground */ 1, /* disney */ 5};
                    if (lobes_of[m.type] > hs->max_lobes) hs->max_lobes = lobes_of[m.type];
                }
                m.ntex = pp.mat.n_textures;
                m.id = pp.mat.id;
                m.table = -1;
                for (int k = 0; k < 4; k++) m.vals[k] = pp.mat.vals[k];
                for (int k = 0; k < 16; k++) {
                    const PtrTexture &t = pp.mat.textures[k];
                    DTex &d = m.tex[k];
                    d.type = t.type;
                    d.image = t.image;
                    if (t.type == 1) { d.v[0] = t.fval; d.v[1] = d.v[2] = 0; }
                    else { d.v[0] = t.spec[0]; d.v[1] = t.spec[1]; d.v[2] = t.spec[2]; }
                    if (t.type == 2 && (t.image < 0 || t.image >= fs->n_images)) return fail(err, PTR_ERR_INVALID, "texture image index out of range");
                    if (t.type < 0 || t.type > 2) { d.type = 0; }
                }
                if (pp.mat.type == PTR_MAT_SUBSURFACE || pp.mat.type == PTR_MAT_KDSUBSURFACE) {
                    /* subsurface.cpp:15-16 (g = vals[2], eta = vals[1]) / kdsubsurface.cpp:33 (g = 0, eta = vals[0]);
                     * the reference builds one table per material, equal inputs give equal tables: shared here */
                    const float g = pp.mat.type == PTR_MAT_SUBSURFACE ? pp.mat.vals[2] : 0.f;
                    const float eta = pp.mat.type == PTR_MAT_SUBSURFACE ? pp.mat.vals[1] : pp.mat.vals[0];
                    for (size_t k = 0; k < hs->tables.size() && m.table < 0; k++)
                        if (hs->tables[k].g == g && hs->tables[k].eta == eta) m.table = (int)k;
                    if (m.table < 0) {
                        m.table = (int)hs->tables.size();
                        hs->tables.emplace_back();
                        hs->tables.back().g = g;
                        hs->tables.back().eta = eta;
                        ptr_build_bssrdf_table(g, eta, &hs->tables.back());
                    }
                }
                pr.material = (int)hs->mats.size();
                hs->mats.push_back(m);
            }
            if (pp.medium.is_valid) {
                medium = (int)hs->media.size();
                hs->media.push_back(conv_medium(pp.medium));
                hs->has_media = true;
            }
        }
        pr.med_inside = medium;
        pr.med_outside = -1;
        hs->prims.push_back(pr);
    }

    /* lights: emissive primitives in insertion order, then the light map (primitive.cpp:205-244) */
    for (size_t i = 0; i < hs->prims.size(); i++) {
        if (!(hs->prims[i].flags & PF_EMISSIVE)) continue;
        DLight l;
        memset(&l, 0, sizeof(l));
        l.type = LT_AREA;
        l.prim = (int)i;
        for (int k = 0; k < 3; k++) l.Lemit[k] = hs->prims[i].L[k];
        memcpy(l.l2w, hs->prims[i].o2w, sizeof(l.l2w));
        memcpy(l.w2l, hs->prims[i].w2o, sizeof(l.w2l));
        hs->prims[i].light = (int)hs->lights.size();
        hs->lights.push_back(l);
    }
    hs->has_env = fs->has_envmap != 0;
    if (hs->has_env) {
        const PtrEnvMap &e = fs->envmap;
        if (e.width <= 0 || e.height <= 0 || e.nu <= 0 || e.nv <= 0 || !e.texels || !e.func || !e.cdf || !e.func_int || !e.marg_cdf)
            return fail(err, PTR_ERR_INVALID, "bad environment map");
        if (!is_affine(e.light_to_world.m) || !is_affine(e.light_to_world.mInv)) return fail(err, PTR_ERR_LIMIT, "projective light transform");
        hs->env_w = e.width; hs->env_h = e.height; hs->env_nu = e.nu; hs->env_nv = e.nv;
        hs->env_texels.assign(e.texels, e.texels + 3 * (size_t)e.width * e.height);
        hs->env_func.assign(e.func, e.func + (size_t)e.nu * e.nv);
        hs->env_cdf.assign(e.cdf, e.cdf + (size_t)(e.nu + 1) * e.nv);
        hs->env_func_int.assign(e.func_int, e.func_int + e.nv);
        hs->env_marg_cdf.assign(e.marg_cdf, e.marg_cdf + e.nv + 1);
        hs->env_marg_int = e.marg_int;
        DLight l;
        memset(&l, 0, sizeof(l));
        l.type = LT_INFINITE;
        l.prim = -1;
        copy34(l.l2w, e.light_to_world.m);
        copy34(l.w2l, e.light_to_world.mInv);
        if (hs->prims.empty()) return fail(err, PTR_ERR_INVALID, "environment map without any primitive");
        l.worldRadius = world_radius(world); /* infinite.cpp:52-56 */
        hs->lights.push_back(l);
    }
    /* distant lights (Aggregator::InsertDistantLight, primitive.cpp:180-189; Init_Distant light.cpp:58-63 with an identity
     * LightToWorld; Distant_Prepare distant.cpp:9-13) */
    if (fs->n_distant < 0 || (fs->n_distant > 0 && !fs->distant)) return fail(err, PTR_ERR_INVALID, "bad distant-light array");
    for (int i = 0; i < fs->n_distant; i++) {
        if (hs->prims.empty()) return fail(err, PTR_ERR_INVALID, "distant light without any primitive");
        const PtrDistantLight &d = fs->distant[i];
        DLight l;
        memset(&l, 0, sizeof(l));
        l.type = LT_DISTANT;
        l.prim = -1;
        for (int k = 0; k < 3; k++) l.Lemit[k] = d.L[k];
        const float len = sqrtf(d.w[0] * d.w[0] + d.w[1] * d.w[1] + d.w[2] * d.w[2]); /* Normalize: v / Length() = v * (1 / len) */
        const float inv = 1.f / len;
        for (int k = 0; k < 3; k++) l.wLight[k] = d.w[k] * inv;
        const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        memcpy(l.l2w, ident, sizeof(ident));
        memcpy(l.w2l, ident, sizeof(ident));
        l.worldRadius = world_radius(world);
        hs->lights.push_back(l);
    }
    /* what the kernels index without checks must be in range before anything is uploaded */
    for (const DPrim &pr : hs->prims) {
        if (pr.type == SH_TOY && pr.toy_id >= 0 && pr.toy_id <= 3 &&
            (pr.root < 0 || pr.geom < 0 || (size_t)pr.root + (size_t)pr.geom > hs->toy_tris.size()))
            return fail(err, PTR_ERR_STATE, "internal: toy triangle range out of bounds");
        if (pr.material >= 0 && hs->mats[pr.material].table >= (int)hs->tables.size())
            return fail(err, PTR_ERR_STATE, "internal: BSSRDF table index out of bounds");
    }
    if (hs->lights.size() > PTR_MAX_LIGHTS) return fail(err, PTR_ERR_LIMIT, "more than 256 lights (Aggregator::lightList[256])");

    /* The top level is the reference's own BVH over primitives (primitive.cpp:537-582,620-642):
     * a few answers depend on the order in which it meets primitives (an SDF hit distance is
     * measured along the normalised direction and can inflate tMax of a non-unit shadow ray,
     * procedural.cpp:238; a rectangle's hit can lie in front of its bounds, rectangle.cpp:8 vs
     * :37-40), and that order follows from the tree.  The tree takes its sort axes from host
     * rand() (util.h:64-66), drawn exactly as Aggregator::Wrap draws: one call per node of each
     * mesh BVH, then of each cloud BVH, then of this one.  It holds tens of primitives; the
     * geometry below it (triangles, particles) lives in the SAH BVHs built above. */
    {
        std::vector<char> mesh_done(hs->meshes.size(), 0), cloud_done(hs->clouds.size(), 0);
        for (const DPrim &pr : hs->prims)
            if (pr.type == SH_MESH && !mesh_done[pr.geom]) { mesh_done[pr.geom] = 1; burn_bvh_rand(hs->meshes[pr.geom].n_tris, 0, 20); }
        for (const DPrim &pr : hs->prims)
            if (pr.type == SH_CLOUD && !cloud_done[pr.geom]) { cloud_done[pr.geom] = 1; burn_bvh_rand(hs->clouds[pr.geom].n, 0, 20); }
        std::vector<TopHandle> th(hs->prims.size());
        for (size_t i = 0; i < hs->prims.size(); i++) {
            for (int a = 0; a < 3; a++) { th[i].pmin[a] = hs->prims[i].bmin[a]; th[i].pmax[a] = hs->prims[i].bmax[a]; }
            th[i].prim = (int32_t)i;
        }
        if (!th.empty()) build_top(hs->top, hs->order, th.data(), (int)th.size(), 0, 20, -1);
        /* The walk only follows left / right / first: renumber the nodes breadth first (the upper levels, which every ray
         * steps through, become top[0 .. TOP_LDS_NODES)) and give every inner node copies of its children's boxes. */
        finish_top_tree(hs->top, hs->order, 0);
        hs->top_ref_nodes = (int)hs->top.size();
        /* ... and behind it the tree most rays walk instead (build_fast_top above; PTR_TOP_TREE=ref: the reference's tree only).
         * Worth it from a handful of primitives on: with two or three the trees are the same. */
        const char *tt = getenv("PTR_TOP_TREE");
        const bool want_fast = !(tt && !strcmp(tt, "ref")) && hs->prims.size() >= 4;
        if (want_fast) {
            std::vector<FastHandle> fh(hs->prims.size());
            bool ok = true;
            for (size_t i = 0; i < hs->prims.size() && ok; i++) {
                ok = fast_top_box(hs->prims[i], &fh[i]);
                fh[i].prim = (int32_t)i;
            }
            if (ok) {
                std::vector<TopNode> fast;
                std::vector<int32_t> forder;
                build_fast_top(fast, forder, fh.data(), (int)fh.size());
                finish_top_tree(fast, forder, (int32_t)hs->top_ref_nodes);
                const int32_t obase = (int32_t)hs->order.size();
                for (TopNode &nd : fast)
                    if (nd.left < 0) nd.first += obase;
                hs->top.insert(hs->top.end(), fast.begin(), fast.end());
                hs->order.insert(hs->order.end(), forder.begin(), forder.end());
            }
        }
    }
    return PTR_OK;
}
