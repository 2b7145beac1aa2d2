/*
 * ptr_host.h — host-side staging of a PtrFlatScene into the flat device tables of dev_scene.h.
 * Replaces the reference's scene bring-up kernels (MakeSceneGPU / Aggregator::Wrap / MakeLights,
 * scene.cpp:891-958, primitive.cpp:620-642) and its three median-split BVH builds with ONE
 * binned-SAH build over triangles, particles and analytic primitives.
 */
#pragma once
#include "../../include/ptr_api.h"
#include "dev_scene.h"
#include <string>
#include <vector>

struct HostImage {
    std::vector<float> rgb;
    int w, h;
};
struct HostMesh {
    int64_t tri_base;
    int n_tris, n_uvs;
    std::vector<float> uv; /* 2*(n_uvs+1): one pad element, mesh.cpp:76 accepts index == nUvs */
};
struct HostCloud {
    std::vector<float> pos;
    int n;
    float radius;
};
struct HostTable { /* BSSRDFTable, bssrdf.h:29-42 */
    int n_rho, n_radius;
    float g, eta; /* what it was built for */
    std::vector<float> rho, radius, profile, rho_eff, profile_cdf;
};
struct HostScene {
    std::vector<Node128> nodes;
    std::vector<NodeQ> qnodes; /* nodes[] quantised (ptr_build.cpp quantise_nodes) */
    std::vector<Item48> items;
    std::vector<DPrim> prims;
    std::vector<DMat> mats;
    std::vector<DMedium> media;
    std::vector<DLight> lights;
    std::vector<HostImage> images;
    std::vector<HostMesh> meshes;
    std::vector<HostCloud> clouds;
    std::vector<HostTable> tables;
    std::vector<DToyTri> toy_tris;
    std::vector<float> tris;
    std::vector<int32_t> uvidx;
    std::vector<TopNode> top;   /* the reference's BVH over primitives: top[0 .. top_ref_nodes); behind it the fast tree (ptr_build.cpp), if built */
    std::vector<int32_t> order; /* their leaf handles: primitive indices */
    int top_ref_nodes = 0;
    std::vector<float> env_texels, env_func, env_cdf, env_func_int, env_marg_cdf;
    float env_marg_int = 0;
    int env_w = 0, env_h = 0, env_nu = 0, env_nv = 0;
    int view_medium = -1, has_env = 0;
    int bvh_depth = 0;
    bool has_media = false;
    int max_lobes = 1; /* the most BxDFs any material of the scene pushes (sizes the lobe array of the SSS shade kernel) */
    int bvh_builder = 0;     /* PTR_BVH_HOST_SAH / PTR_BVH_DEVICE_LBVH: who built the geometry trees */
    double bvh_build_ms = 0; /* all geometry trees: host wall time (SAH) or device time (LBVH) */
    /* surface-area estimate of the traversal cost of the largest geometry tree, per ray that enters its root box:
     * child boxes tested, leaf items tested, leaves entered */
    double sah_boxes = 0, sah_items = 0, sah_leaves = 0;
    size_t sah_of_items = 0;
};

/* light-map tables of one environment map (rt_envmap.hip), host copies */
struct HostEnvTables {
    int width = 0, height = 0, nu = 0, nv = 0;
    std::vector<float> texels, func, cdf, func_int, marg_cdf;
    float marg_int = 0;
    double build_ms = 0;
};
int ptr_gpu_build_envmap(const float *rgb, int width, int height, const float scale[3], HostEnvTables *out, double *ms, std::string *err);
/* geometry BVH of one mesh / particle cloud on the current device (rt_build.hip) */
int ptr_gpu_build_geom(const float *boxes6, const Item48 *raw, size_t n_items, std::vector<Node128> *nodes, std::vector<Item48> *items, int32_t *root,
                       float *bound6, int *depth, double *ms, std::string *err);
/* which builder ptr_flatten uses (ptr_scene_build_mode / environment PTR_BVH_BUILD=device) */
int ptr_bvh_build_mode_get();
void ptr_bvh_build_mode_set(int mode);

/* returns PTR_OK or an error code with *err filled */
/* photon-beam-diffusion profile table of a subsurface material (ptr_tables.cpp) */
void ptr_build_bssrdf_table(float g, float eta, HostTable *t);
/* triangle records of an origami toy (ptr_toys.cpp); returns how many were appended */
int ptr_build_toy(int id, const float *half, std::vector<DToyTri> *out);
int ptr_flatten(const PtrFlatScene *fs, HostScene *out, std::string *err);
