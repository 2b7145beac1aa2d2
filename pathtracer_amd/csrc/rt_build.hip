/*
 * rt_build.hip — geometry BVH build on the device.
 *
 * Replaces, for one mesh or particle cloud, the host binned-SAH build + collapse of ptr_build.cpp (itself the
 * replacement of the reference's _CreateBVH, src/core/primitive.cpp:537-582, which takes 14 s for 7 M triangles on the
 * author's machine, and of the per-shape builds mesh.cpp:643-671 / particlecloud.cpp:310-337).  Same output tables: 4-wide
 * child-major Node128 nodes and pre-gathered Item48 leaf records, leaves of up to BVH_MAX_LEAF items.
 *
 *   1. k_item_bounds   centroid bounds of all items                       (block reduction + 6 ordered-int atomics)
 *   2. k_morton        30-bit Morton code of each centroid, key = code << 32 | item index (unique keys)
 *   3. hipcub radix sort of the 64-bit keys                               (the only library call: rocPRIM's onesweep sort)
 *   4. k_hierarchy     Karras 2012: one thread per inner node of the binary radix tree finds its range and split
 *                      from the common prefixes of neighbouring keys; a node covers a CONTIGUOUS range of sorted items
 *   5. k_refit         leaves to root, the second arrival at a node (atomic flag) unions its children's boxes
 *   6. k_collapse      breadth first, one launch per level of the 4-wide tree: a work item (binary node -> wide node)
 *                      opens the child of largest area until four slots are taken (the host collapser's rule); a
 *                      subtree of at most BVH_MAX_LEAF items becomes a leaf when the host builder's SAH rule says so
 *                      (makes_leaf) - its items are contiguous, so the leaf ref is (first, count) - anything else
 *                      gets a wide node of its own and goes to the next level
 *   7. k_gather_items  Item48 records in sorted order
 *
 * Every step is a streaming or gather kernel over N items: HBM-bound; the sort dominates.  The tree is a linear BVH:
 * cheaper to build than SAH by orders of magnitude, somewhat more expensive to traverse (bench.py reports both).
 * Results of a trace do not depend on which builder made the tree, except for hits at bit-equal t (dev_isect.h).
 */
#include "rt_common.h"
#include "ptr_host.h"
#include <hipcub/hipcub.hpp>
#include <cfloat>
#include <vector>

struct GBox {
    float mn[3], mx[3];
};

__device__ __forceinline__ unsigned ord_of(float f) { /* float -> unsigned with the same order */
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float float_of(unsigned o) {
    unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* 1. bounds of the item centroids (cen[0..2] min, cen[3..5] max) and of the items themselves (all[0..5]), as ordered ints */
__global__ void __launch_bounds__(TPB) k_item_bounds(const GBox *boxes, int n, unsigned *cen, unsigned *all) {
    float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    float amn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, amx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
        const GBox b = boxes[i];
        for (int a = 0; a < 3; a++) {
            const float c = 0.5f * (b.mn[a] + b.mx[a]);
            cmn[a] = fminf(cmn[a], c); cmx[a] = fmaxf(cmx[a], c);
            amn[a] = fminf(amn[a], b.mn[a]); amx[a] = fmaxf(amx[a], b.mx[a]);
        }
    }
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) {
            cmn[a] = fminf(cmn[a], __shfl_xor(cmn[a], o, 64)); cmx[a] = fmaxf(cmx[a], __shfl_xor(cmx[a], o, 64));
            amn[a] = fminf(amn[a], __shfl_xor(amn[a], o, 64)); amx[a] = fmaxf(amx[a], __shfl_xor(amx[a], o, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        for (int a = 0; a < 3; a++) {
            atomicMin(&cen[a], ord_of(cmn[a])); atomicMax(&cen[3 + a], ord_of(cmx[a]));
            atomicMin(&all[a], ord_of(amn[a])); atomicMax(&all[3 + a], ord_of(amx[a]));
        }
    }
}

__device__ __forceinline__ unsigned expand10(unsigned v) { /* 10 bits -> every third bit */
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
/* 2. */
__global__ void __launch_bounds__(TPB) k_morton(const GBox *boxes, int n, float ox, float oy, float oz, float sx, float sy, float sz, unsigned long long *keys) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const GBox b = boxes[i];
    const float cx = (0.5f * (b.mn[0] + b.mx[0]) - ox) * sx, cy = (0.5f * (b.mn[1] + b.mx[1]) - oy) * sy, cz = (0.5f * (b.mn[2] + b.mx[2]) - oz) * sz;
    const unsigned x = (unsigned)fminf(fmaxf(cx, 0.f), 1023.f), y = (unsigned)fminf(fmaxf(cy, 0.f), 1023.f), z = (unsigned)fminf(fmaxf(cz, 0.f), 1023.f);
    const unsigned code = (expand10(x) << 2) | (expand10(y) << 1) | expand10(z);
    keys[i] = ((unsigned long long)code << 32) | (unsigned)i;
}

/* binary radix tree over n sorted keys: inner nodes 0 .. n-2 (0 = root); a child ref >= 0 is an inner node, < 0 is leaf ~ref */
struct BinTree {
    int *left, *right, *parent; /* parent[inner], leaf_parent[leaf] */
    int *leaf_parent;
    int *first, *last;          /* range of sorted items an inner node covers */
    GBox *box;                  /* of inner nodes */
    int *flag;                  /* refit arrival counters */
};
__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}
/* 4. Karras, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees", HPG 2012 */
__global__ void __launch_bounds__(TPB) k_hierarchy(const unsigned long long *keys, int n, BinTree t) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int s = lmax >> 1; s >= 1; s >>= 1)
        if (delta(keys, n, i, i + (l + s) * d) > dmin) l += s;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0, tt = l;
    do {
        tt = (tt + 1) >> 1;
        if (delta(keys, n, i, i + (s + tt) * d) > dnode) s += tt;
    } while (tt > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int lref = lo == gamma ? ~gamma : gamma;
    const int rref = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    t.left[i] = lref;
    t.right[i] = rref;
    t.first[i] = lo;
    t.last[i] = hi;
    if (lref >= 0) t.parent[lref] = i; else t.leaf_parent[~lref] = i;
    if (rref >= 0) t.parent[rref] = i; else t.leaf_parent[~rref] = i;
    if (i == 0) t.parent[0] = -1;
}

__device__ __forceinline__ GBox box_union(const GBox &a, const GBox &b) {
    GBox r;
    for (int k = 0; k < 3; k++) { r.mn[k] = fminf(a.mn[k], b.mn[k]); r.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
    return r;
}
__device__ __forceinline__ GBox ref_box(const BinTree &t, const GBox *boxes, const unsigned long long *keys, int ref) {
    return ref >= 0 ? t.box[ref] : boxes[(unsigned)(keys[~ref] & 0xffffffffu)];
}
/* 5. one thread per leaf climbs; the first arrival at a node stops, the second (whose sibling is done and visible: the
 * arrival counter is a device-scope atomic and the boxes are read behind a fence) computes the node and goes on */
__global__ void __launch_bounds__(TPB) k_refit(const GBox *boxes, const unsigned long long *keys, int n, BinTree t) {
    const int leaf = blockIdx.x * TPB + threadIdx.x;
    if (leaf >= n) return;
    int node = t.leaf_parent[leaf];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&t.flag[node], 1) == 0) return;
        __threadfence();
        /* a child that is an inner node was written by another workgroup in this launch: read it past this CU's L1 */
        GBox la = ref_box(t, boxes, keys, t.left[node] >= 0 ? -1 : t.left[node]), lb = ref_box(t, boxes, keys, t.right[node] >= 0 ? -1 : t.right[node]);
        if (t.left[node] >= 0) { const float *p = (const float *)&t.box[t.left[node]]; for (int k = 0; k < 6; k++) ((float *)&la)[k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (t.right[node] >= 0) { const float *p = (const float *)&t.box[t.right[node]]; for (int k = 0; k < 6; k++) ((float *)&lb)[k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        const GBox u = box_union(la, lb);
        float *q = (float *)&t.box[node];
        for (int k = 0; k < 6; k++) __hip_atomic_store(q + k, ((const float *)&u)[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        node = t.parent[node];
    }
}

__device__ __forceinline__ float half_area(const GBox &b) {
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ int ref_count(const BinTree &t, int ref) { return ref >= 0 ? t.last[ref] - t.first[ref] + 1 : 1; }
__device__ __forceinline__ int ref_first(const BinTree &t, int ref) { return ref >= 0 ? t.first[ref] : ~ref; }

/* Does the subtree under ref become one leaf?  Up to two items: always.  Three or four: by the host builder's rule
 * (ptr_build.cpp): a leaf unless testing the two halves separately is expected to be cheaper,
 * n > 1 + (A_l n_l + A_r n_r) / A - which it is for items that overlap little, not for a cluster of particle spheres */
__device__ __forceinline__ bool makes_leaf(const BinTree &t, const GBox *boxes, const unsigned long long *keys, int ref, const GBox &box) {
    if (ref < 0) return true;
    const int n = ref_count(t, ref);
    if (n > BVH_MAX_LEAF) return false;
    if (n <= 2) return true;
    const int l = t.left[ref], r = t.right[ref];
    const float a = half_area(box);
    if (!(a > 0.f)) return true;
    const float split = 1.f + (half_area(ref_box(t, boxes, keys, l)) * ref_count(t, l) + half_area(ref_box(t, boxes, keys, r)) * ref_count(t, r)) / a;
    return (float)n <= split;
}

/* 6. one level of the 4-wide tree.  work[k] = (binary inner node, wide node index) */
__global__ void __launch_bounds__(TPB) k_collapse(const GBox *boxes, const unsigned long long *keys, BinTree t, const int2 *work, int n_work, int2 *next, int *n_next,
                                                  int *n_wide, Node128 *out, int node_base, unsigned item_base) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k >= n_work) return;
    const int bin = work[k].x, me = work[k].y;
    int refs[4];
    GBox bx[4];
    bool leaf[4];
    int ns = 2;
    refs[0] = t.left[bin];
    refs[1] = t.right[bin];
    for (int i = 0; i < 2; i++) {
        bx[i] = ref_box(t, boxes, keys, refs[i]);
        leaf[i] = makes_leaf(t, boxes, keys, refs[i], bx[i]);
    }
    while (ns < 4) { /* open the child of largest area that does not become a leaf */
        int pick = -1;
        float area = -1.f;
        for (int i = 0; i < ns; i++)
            if (!leaf[i]) {
                const float a = half_area(bx[i]);
                if (a > area) { area = a; pick = i; }
            }
        if (pick < 0) break;
        const int l = t.left[refs[pick]], r = t.right[refs[pick]];
        refs[pick] = l;
        bx[pick] = ref_box(t, boxes, keys, l);
        leaf[pick] = makes_leaf(t, boxes, keys, l, bx[pick]);
        refs[ns] = r;
        bx[ns] = ref_box(t, boxes, keys, r);
        leaf[ns] = makes_leaf(t, boxes, keys, r, bx[ns]);
        ns++;
    }
    Node128 nd;
    for (int i = 0; i < 4; i++) {
        if (i < ns) {
            const int cnt = ref_count(t, refs[i]);
            int ref;
            if (leaf[i]) {
                const unsigned packed = (item_base + (unsigned)ref_first(t, refs[i])) | ((unsigned)(cnt - 1) << 27);
                ref = -(int)packed - 1;
            } else {
                const int w = atomicAdd(n_wide, 1);
                const int q = atomicAdd(n_next, 1);
                next[q] = make_int2(refs[i], w);
                ref = node_base + w;
            }
            for (int a = 0; a < 3; a++) { nd.c[i].mn[a] = bx[i].mn[a]; nd.c[i].mx[a] = bx[i].mx[a]; }
            nd.c[i].ref = ref;
            nd.c[i].pad = 0;
        } else {
            for (int a = 0; a < 3; a++) { nd.c[i].mn[a] = 0; nd.c[i].mx[a] = 0; }
            nd.c[i].ref = BVH_NO_CHILD;
            nd.c[i].pad = 0;
        }
    }
    out[me] = nd;
}

/* 7. */
__global__ void __launch_bounds__(TPB) k_gather_items(const Item48 *raw, const unsigned long long *keys, int n, Item48 *out) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const float4 *src = reinterpret_cast<const float4 *>(raw + (unsigned)(keys[i] & 0xffffffffu));
    float4 *dst = reinterpret_cast<float4 *>(out + i);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

#define GCHK(expr)                                                                     \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) { if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e_); rc = PTR_ERR_HIP; goto done; } \
    } while (0)

/* boxes / raw: n host records.  Appends the 4-wide nodes to *nodes (their refs are global: node_base = nodes->size() at entry)
 * and the items, in leaf order, to *items.  *root: ref of the tree, *bound: its box, *depth: levels of the 4-wide tree,
 * *ms: device time of steps 1-7 (without the copies in and out). */
int ptr_gpu_build_geom(const float *boxes6, const Item48 *raw, size_t n_items, std::vector<Node128> *nodes, std::vector<Item48> *items, int32_t *root,
                       float *bound6, int *depth, double *ms, std::string *err) {
    int rc = PTR_OK;
    const int n = (int)n_items;
    const int node_base = (int)nodes->size();
    const unsigned item_base = (unsigned)items->size();
    *depth = 0;
    if (ms) *ms = 0;
    if (n == 0) return PTR_OK;
    GBox *d_boxes = nullptr;
    Item48 *d_raw = nullptr, *d_items = nullptr;
    unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
    unsigned *d_bounds = nullptr;
    void *d_temp = nullptr;
    int *d_ints = nullptr;
    GBox *d_nbox = nullptr;
    Node128 *d_out = nullptr;
    int2 *d_work[2] = {nullptr, nullptr};
    int *d_counts = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    size_t temp_bytes = 0;
    unsigned hb[12];
    const unsigned blocks = (unsigned)((n + TPB - 1) / TPB);
    int n_wide = 0;
    {
        GCHK(hipMalloc((void **)&d_boxes, sizeof(GBox) * n));
        GCHK(hipMalloc((void **)&d_raw, sizeof(Item48) * n));
        GCHK(hipMalloc((void **)&d_items, sizeof(Item48) * n));
        GCHK(hipMalloc((void **)&d_keys, 8 * (size_t)n));
        GCHK(hipMalloc((void **)&d_keys2, 8 * (size_t)n));
        GCHK(hipMalloc((void **)&d_bounds, 12 * sizeof(unsigned)));
        GCHK(hipMemcpy(d_boxes, boxes6, sizeof(GBox) * n, hipMemcpyHostToDevice));
        GCHK(hipMemcpy(d_raw, raw, sizeof(Item48) * n, hipMemcpyHostToDevice));
        GCHK(hipEventCreate(&e0));
        GCHK(hipEventCreate(&e1));
        GCHK(hipEventRecord(e0, 0));
        /* 1 */
        for (int a = 0; a < 3; a++) { hb[a] = 0xffffffffu; hb[3 + a] = 0; hb[6 + a] = 0xffffffffu; hb[9 + a] = 0; }
        GCHK(hipMemcpy(d_bounds, hb, sizeof(hb), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_item_bounds, dim3(blocks < 1024 ? blocks : 1024), dim3(TPB), 0, 0, d_boxes, n, d_bounds, d_bounds + 6);
        GCHK(hipMemcpy(hb, d_bounds, sizeof(hb), hipMemcpyDeviceToHost));
        float cmn[3], cmx[3];
        for (int a = 0; a < 3; a++) { cmn[a] = float_of(hb[a]); cmx[a] = float_of(hb[3 + a]); bound6[a] = float_of(hb[6 + a]); bound6[3 + a] = float_of(hb[9 + a]); }
        if (n == 1) { /* a single item: the tree is one leaf */
            hipLaunchKernelGGL(k_morton, dim3(1), dim3(TPB), 0, 0, d_boxes, n, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, d_keys);
            hipLaunchKernelGGL(k_gather_items, dim3(1), dim3(TPB), 0, 0, d_raw, d_keys, n, d_items);
            *root = -(int32_t)(item_base) - 1;
        } else {
            /* 2, 3 */
            float sc3[3];
            for (int a = 0; a < 3; a++) sc3[a] = cmx[a] > cmn[a] ? 1024.f / (cmx[a] - cmn[a]) : 0.f;
            hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(TPB), 0, 0, d_boxes, n, cmn[0], cmn[1], cmn[2], sc3[0], sc3[1], sc3[2], d_keys);
            GCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, d_keys, d_keys2, n, 0, 62, 0));
            GCHK(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 16));
            GCHK(hipcub::DeviceRadixSort::SortKeys(d_temp, temp_bytes, d_keys, d_keys2, n, 0, 62, 0));
            /* 4, 5 */
            GCHK(hipMalloc((void **)&d_ints, sizeof(int) * 7 * (size_t)n));
            GCHK(hipMalloc((void **)&d_nbox, sizeof(GBox) * n));
            BinTree t;
            t.left = d_ints; t.right = d_ints + n; t.parent = d_ints + 2 * (size_t)n; t.leaf_parent = d_ints + 3 * (size_t)n;
            t.first = d_ints + 4 * (size_t)n; t.last = d_ints + 5 * (size_t)n; t.flag = d_ints + 6 * (size_t)n;
            t.box = d_nbox;
            GCHK(hipMemsetAsync(t.flag, 0, sizeof(int) * n, 0));
            hipLaunchKernelGGL(k_hierarchy, dim3(blocks), dim3(TPB), 0, 0, d_keys2, n, t);
            hipLaunchKernelGGL(k_refit, dim3(blocks), dim3(TPB), 0, 0, d_boxes, d_keys2, n, t);
            /* 6 */
            GCHK(hipMalloc((void **)&d_out, sizeof(Node128) * n)); /* at most n - 1 wide nodes */
            GCHK(hipMalloc((void **)&d_work[0], sizeof(int2) * n));
            GCHK(hipMalloc((void **)&d_work[1], sizeof(int2) * n));
            GCHK(hipMalloc((void **)&d_counts, 3 * sizeof(int)));
            if (n <= 2) {
                *root = -(int32_t)(item_base | ((unsigned)(n - 1) << 27)) - 1; /* the whole geometry is one leaf */
            } else {
                const int2 first_work = make_int2(0, 0);
                int counts[3] = {0, 0, 1}; /* next-queue sizes (ping, pong), wide nodes so far */
                GCHK(hipMemcpy(d_work[0], &first_work, sizeof(int2), hipMemcpyHostToDevice));
                GCHK(hipMemcpy(d_counts, counts, sizeof(counts), hipMemcpyHostToDevice));
                int n_work = 1, cur = 0;
                while (n_work > 0) {
                    (*depth)++;
                    GCHK(hipMemsetAsync(d_counts + (cur ^ 1), 0, sizeof(int), 0));
                    hipLaunchKernelGGL(k_collapse, dim3((unsigned)((n_work + TPB - 1) / TPB)), dim3(TPB), 0, 0, d_boxes, d_keys2, t, (const int2 *)d_work[cur], n_work,
                                       d_work[cur ^ 1], d_counts + (cur ^ 1), d_counts + 2, d_out, node_base, item_base);
                    GCHK(hipMemcpy(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost));
                    n_work = counts[cur ^ 1];
                    cur ^= 1;
                    if (*depth > 64) { if (err) *err = "device BVH build: the 4-wide tree is deeper than 64 levels"; rc = PTR_ERR_LIMIT; goto done; }
                }
                n_wide = counts[2];
                *root = node_base;
            }
            /* 7 */
            hipLaunchKernelGGL(k_gather_items, dim3(blocks), dim3(TPB), 0, 0, d_raw, d_keys2, n, d_items);
        }
        GCHK(hipGetLastError());
        GCHK(hipEventRecord(e1, 0));
        GCHK(hipEventSynchronize(e1));
        float fms = 0;
        GCHK(hipEventElapsedTime(&fms, e0, e1));
        if (ms) *ms = fms;
        nodes->resize((size_t)node_base + n_wide);
        if (n_wide) GCHK(hipMemcpy(nodes->data() + node_base, d_out, sizeof(Node128) * n_wide, hipMemcpyDeviceToHost));
        items->resize((size_t)item_base + n);
        GCHK(hipMemcpy(items->data() + item_base, d_items, sizeof(Item48) * n, hipMemcpyDeviceToHost));
    }
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    void *frees[] = {d_boxes, d_raw, d_items, d_keys, d_keys2, d_bounds, d_temp, d_ints, d_nbox, d_out, d_work[0], d_work[1], d_counts};
    for (void *p : frees)
        if (p) (void)hipFree(p);
    return rc;
}
