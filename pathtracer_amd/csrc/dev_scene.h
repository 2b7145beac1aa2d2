/*
 * dev_scene.h — flat device-resident scene tables (what the reference keeps as a pointer-linked
 * object graph in managed memory: Aggregator/Primitive/Shape/Material/Light, primitive.h:13-130).
 * Shared by the host flattening code (ptr_scene.cpp) and the kernels.
 */
#pragma once
#include <cstdint>

enum { SH_SPHERE = 0, SH_RECT = 1, SH_DISK = 2, SH_BOX = 3, SH_MESH = 4, SH_CLOUD = 5, SH_SPHERE_PROC = 6, SH_BOX_PROC = 7, SH_TOY = 8 };
enum { PF_FLIP_N = 1, PF_REVERSE = 2, PF_EMISSIVE = 4 };
enum { ITEM_TRI = 0, ITEM_PARTICLE = 1, ITEM_PRIM = 2 };
enum { LT_AREA = 1, LT_DISTANT = 2, LT_INFINITE = 3 };

/* Inner node of a geometry BVH: FOUR children per node, 128 B = one L2 line, child-major: child i is the
 * 32-byte record c[i] = {box min xyz, box max xyz, ref, -}.  ref >= 0 inner node index, < 0 leaf
 * `-(first | (n-1) << 27) - 1` over items[], BVH_NO_CHILD = empty slot.  Made by collapsing the
 * binned-SAH binary tree (ptr_build.cpp): half as many dependent node fetches per ray as a binary
 * tree, and every byte of the fetched line is used.  Child-major so that the four lanes that share a
 * ray in the quad tracer (dev_qtrace.h) each fetch their child with two 16-byte loads; the one-ray-per-lane
 * walkers read the whole line (dev_isect.h bvh4_visit). */
struct Node128 {
    struct Child {
        float mn[3], mx[3];
        int32_t ref;
        int32_t pad;
    } c[4];
};
/* The same node for the one-ray-per-lane walkers of the persistent tracer, 64 B: the children's boxes quantised to 8 bits per
 * plane inside the union of the four (plane = o + q * 2^(e - 127), one fused multiply-add; the builder rounds minima down and
 * maxima up against exactly that expression, ptr_build.cpp quantise_nodes).  A lane's fetch of a node is per-lane 16-byte
 * requests to the L1, which serves about one request per clock and CU: the tracer's counters show it saturated,
 * so a visit costs what it asks for - four requests here, eight for a Node128.  The decoded boxes contain the exact ones and
 * the slab test is monotonic in the box, so a ray enters every child it entered before: same candidates, same closest hit. */
#ifndef PT_QNODES
#define PT_QNODES 0 /* make VARIANT=q EXTRA=-DPT_QNODES=1: built, exact (films identical), measured equal to Node128 on every config - off */
#endif
struct NodeQ {
    float o[3];      /* min corner of the union of the children's boxes */
    uint32_t e;      /* biased exponents of the axis scales: ex | ey << 8 | ez << 16 */
    int32_t ref[4];  /* as Node128::Child::ref */
    uint32_t q[6];   /* min x, min y, min z, max x, max y, max z: byte c belongs to child c */
    uint32_t pad[2];
};
/* Node of the top-level BVH over primitives: the reference's own tree (primitive.cpp:537-582),
 * kept because a few results depend on its visiting order.  Leaf: left = -1, handles
 * order[first .. first+n). */
struct TopNode { /* 96 B = 6 float4; each tree's nodes are numbered breadth first (ptr_build.cpp), node 0 is the reference tree's root */
    float bmin[3];
    int32_t left;
    float bmax[3];
    int32_t right;
    int32_t first, n;
    int32_t pad[2]; /* inner node: the primitive of the left / right child if that child is a one-primitive leaf, else -1 (ptr_build.cpp) */
    /* copies of the two children's boxes (inner nodes): one step of the persistent tracer - this node's box, then both
     * children's - reads one record instead of three (dev_trace.h) */
    float lmin[3], lmax[3], rmin[3], rmax[3];
};
#ifndef TOP_LDS_NODES
#define TOP_LDS_NODES 64 /* top[0 .. 64) = the upper six levels, kept in LDS by the persistent tracer */
#endif
#define TOP_LDS_ROW 7     /* float4 per staged node: 96 B + 16 B of padding against bank conflicts */
#define TOP_STACK 48
#define BVH_NO_CHILD INT32_MIN
#define BVH_NODE_PREFIX 64 /* nodes[0 .. 64): the upper levels of all geometry trees, breadth first (ptr_build.cpp) */
#define BVH_MAX_LEAF 4
#define BVH_STACK 96 /* entries per lane: up to 3 pushes per level of a 4-wide tree */

/* Leaf record, 48 B (3 x float4), pre-gathered so a leaf test needs no indirection:
 *  TRI      : q0 = (p0, prim)  q1 = (p1, tri)       q2 = (p2, ITEM_TRI)
 *  PARTICLE : q0 = (c,  prim)  q1 = (r,0,0, index)  q2 = (0,0,0, ITEM_PARTICLE)
 *  PRIM     : q0 = (bmin,prim) q1 = (bmax, 0)       q2 = (0,0,0, ITEM_PRIM)   (ints stored bitwise) */
struct Item48 {
    float q[12];
};

struct DPrim {
    int32_t type, flags, geom, material;
    int32_t light, med_inside, med_outside, toy_id;
    float o2w[12], w2o[12]; /* rows 0..2 of ObjectToWorld.m / WorldToObject.m */
    float bmin[3], bmax[3]; /* Shape::GetBounds() in world space */
    float p[8];             /* sphere: radius,zMin,zMax,thetaMin,thetaMax,phiMax | rect: sizex,sizey |
                               disk: height,radius,innerRadius,phiMax | box: length xyz |
                               toy: bounds pMin xyz, pMax xyz | cloud: radius */
    float L[3];
    float area; /* Shape::Area() (unused for meshes: their sampling works per triangle) */
    int32_t root; /* mesh / cloud: root ref of its BVH (node index, or a leaf ref if < 0) */
};

struct DTex {
    int32_t type; /* 0 spectrum, 1 float, 2 image */
    float v[3];   /* spectrum, or (fval, -, -) */
    int32_t image;
};
struct DMat {
    int32_t type, ntex, id, table;
    float vals[4];
    DTex tex[16];
};
struct DMedium {
    float sigma_a[3], sigma_s[3], sigma_t[3];
    float g;
};
struct DLight {
    int32_t type, prim;
    float Lemit[3];
    float l2w[12], w2l[12];
    float worldRadius;
    float wLight[3]; /* LT_DISTANT: Normalize(LightToWorld(w)), light.cpp:58-63 */
};
struct DImage {
    const float *rgb;
    int32_t width, height;
};
struct DMesh {
    int64_t tri_base; /* into tris (9 floats each) and uvidx (3 ints each) */
    int32_t n_tris, n_uvs;
    const float *uv;  /* 2*n_uvs or null */
};
struct DCloud {
    const float *pos; /* 3*n */
    int32_t n;
    float radius;
};
/* One triangle of an origami signed-distance toy (procedural.cpp:402-574) with everything
 * SDF::Triangle derives from its corners alone, precomputed on the host (ptr_toys.cpp): 144 B */
struct DToyTri {
    float a[3], b[3], c[3];
    float ba[3], cb[3], ac[3], nor[3];
    float e0[3], e1[3], e2[3]; /* Cross(ba, nor), Cross(cb, nor), Cross(ac, nor) */
    float d2ba, d2cb, d2ac, d2nor;
    float id, pad;
};
struct DTable { /* BSSRDFTable, bssrdf.h:29-42 (100 x 64) */
    const float *rhoSamples, *radiusSamples, *profile, *rhoEff, *profileCDF;
    int32_t nRho, nRadius;
};

struct DScene {
    const Node128 *nodes;
    const NodeQ *qnodes; /* nodes[], quantised (same numbering) */
    const Item48 *items;
    const DPrim *prims;
    const DMat *mats;
    const DMedium *media;
    const DLight *lights;
    const DImage *images;
    const DMesh *meshes;
    const DCloud *clouds;
    const DTable *tables;
    const DToyTri *toy_tris; /* origami toys: DPrim::root = first record, DPrim::geom = count */
    const float *tris;     /* 9 floats / triangle, original order, world space */
    const int32_t *uvidx;  /* 3 ints / triangle */
    const TopNode *top;    /* the reference's BVH over primitives, root = 0 */
    const int32_t *order;  /* its leaf handles: primitive indices */
    /* environment map */
    const float *env_texels, *env_func, *env_cdf, *env_func_int, *env_marg_cdf;
    /* the same CDFs as 16-ary search trees (dev_light.h find_interval_tree; built at upload, null when the map is too large):
     * env_cdf_tree = one tree of env_tree_u_nodes nodes per row, env_marg_tree = one tree */
    const float *env_cdf_tree, *env_marg_tree;
    int32_t env_tree_u_levels, env_tree_u_nodes, env_tree_v_levels, env_tree_pad;
    float env_marg_int;
    int32_t env_w, env_h, env_nu, env_nv;
    int32_t n_prims, n_lights, n_nodes, n_items;
    int32_t view_medium, has_env;
    int32_t n_top;      /* nodes of the reference's tree, top[0 .. n_top) */
    int32_t top_fast;   /* root of the fast tree over the same primitives (ptr_build.cpp build_fast_top), top[top_fast .. top_fast + n_top_fast); 0: none */
    int32_t n_top_fast;
    int32_t trace_waves; /* which build of the persistent tracer the scene's launches use (rt_trace.hip): PT_WAVES or PT_WAVES_WIDE */
    int32_t has_proc;    /* some primitive is an SDF shape (SH_SPHERE_PROC / SH_BOX_PROC / SH_TOY): the tracer builds with their code (rt_trace.hip) */
};

struct DCamera {
    float position[3], lower_left[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    float lens_radius;
};
