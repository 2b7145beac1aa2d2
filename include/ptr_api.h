/*
 * ptr_api.h — C-ABI of the MI355X-native per-ray hot path ("ptr" = path-tracer runtime).
 *
 * This is the drop-in boundary for felpzOliveira/PathTracer's render path.  The
 * reference has no FFI: its host code (src/main.cpp render(), src/core/scene.cpp)
 * launches CUDA kernels directly on managed memory.  Every entry point below names
 * the reference launch site / host function it replaces (paths relative to the
 * reference tree).  Signatures use plain pointers and sizes only.
 *
 * Data crossing the boundary is a *flat scene*: the reference's POD descriptors
 * (src/core/scene.h:11-86) with every pointer-to-object replaced by an index or a
 * plain array.  The library owns everything it uploads; the caller keeps ownership
 * of every array it passes in (they may be freed after ptr_scene_upload returns).
 */
#ifndef PTR_API_H
#define PTR_API_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTR_OK 0
#define PTR_ERR_INVALID 1   /* bad argument / scene description            */
#define PTR_ERR_HIP 2       /* a HIP runtime call failed                   */
#define PTR_ERR_STATE 3     /* call order violated (e.g. render w/o scene) */
#define PTR_ERR_LIMIT 4     /* a documented limit was exceeded             */

#define PTR_MAX_TEXTURES 16 /* MaterialDescriptor::textures[16], scene.h:63 */
#define PTR_MAX_LIGHTS 256  /* Aggregator::lightList[256], primitive.h:93   */

/* ShapeType, src/core/shape.h:8-11 (same numeric values) */
enum {
    PTR_SHAPE_SPHERE = 0,
    PTR_SHAPE_RECTANGLE = 1,
    PTR_SHAPE_DISK = 2,
    PTR_SHAPE_BOX = 3, /* dead in the reference: MakeShape builds BOX_PROCEDURAL, scene.cpp:746 */
    PTR_SHAPE_MESH = 4,
    PTR_SHAPE_PARTICLECLOUD = 5,
    PTR_SHAPE_SPHERE_PROCEDURAL = 6,
    PTR_SHAPE_BOX_PROCEDURAL = 7,
    PTR_SHAPE_COMPONENT_PROCEDURAL = 8
};

/* MaterialType, src/core/material.h:250-254 (same numeric values) */
enum {
    PTR_MAT_NONE = 0,
    PTR_MAT_MATTE = 1,
    PTR_MAT_MIRROR = 2,
    PTR_MAT_GLASS = 3,
    PTR_MAT_METAL = 4,
    PTR_MAT_TRANSLUCENT = 5,
    PTR_MAT_PLASTIC = 6,
    PTR_MAT_UBER = 7,
    PTR_MAT_SUBSURFACE = 8,
    PTR_MAT_KDSUBSURFACE = 9,
    PTR_MAT_PLAYGROUND = 10,
    /* The reference's five Disney lobes (reflection.h:53-57,90-129; reflection.cpp:232-295, 504-521), which no material of
     * the reference instantiates, assembled the way PBRT-v3's DisneyMaterial assembles its own.  Texture slots: 0 colour
     * (Spectrum), 1 roughness, 2 sheen, 3 sheenTint, 4 clearcoat, 5 clearcoatGloss, 6 flatness (Floats). */
    PTR_MAT_DISNEY = 11
};

/* Integrator selector: the reference picks one by editing src/main.cpp:444-447 */
enum {
    PTR_LI_DIRECT = 0,      /* Li_Direct      main.cpp:72  */
    PTR_LI_PATH = 1,        /* Li_Path        main.cpp:362 */
    PTR_LI_PATHSAMPLED = 2, /* Li_PathSampled main.cpp:251 */
    PTR_LI_VOLPATH = 3      /* Li_VolPath     main.cpp:105 */
};

/* Transform{Matrix4x4 m, mInv}, src/core/transform.h:8-49; row-major m[r*4+c] */
typedef struct {
    float m[16];
    float mInv[16];
} PtrTransform;

/* TextureDescriptor, scene.h:52-57: type 0 = constant Spectrum, 1 = constant Float,
 * 2 = image (index into PtrFlatScene::images) */
typedef struct {
    int32_t type;
    float spec[3];
    float fval;
    int32_t image;
} PtrTexture;

/* MaterialDescriptor, scene.h:59-66.  Slot order per material type is the one
 * MakeMaterial reads (scene.cpp:801-889). */
typedef struct {
    int32_t type; /* PTR_MAT_* */
    int32_t is_emissive;
    int32_t n_textures; /* slots filled (specs_taken for PlayGround) */
    int32_t id;
    PtrTexture textures[PTR_MAX_TEXTURES];
    float vals[16];
} PtrMaterial;

/* MediumDescriptor, scene.h:68-72 */
typedef struct {
    float sigma_a[3];
    float sigma_s[3];
    float g;
    int32_t is_valid;
} PtrMedium;

/* PrimitiveDescriptor, scene.h:74-86, with the per-shape descriptors folded into
 * one parameter block:
 *   SPHERE / SPHERE_PROCEDURAL : params[0]=radius
 *   RECTANGLE                  : params[0]=sizex, [1]=sizey
 *   DISK                       : params[0]=height, [1]=radius, [2]=innerRadius, [3]=phiMax(deg)
 *   BOX_PROCEDURAL             : params[0..2]=sizex,sizey,sizez
 *   COMPONENT_PROCEDURAL       : params[0..2]=bound.pMin, [3..5]=bound.pMax; geom=toy id
 *   MESH                       : geom = index into meshes (to_world unused: the mesh carries it)
 *   PARTICLECLOUD              : geom = index into clouds
 */
typedef struct {
    int32_t shape_type;
    int32_t reverse_orientation;
    int32_t geom;
    int32_t no_mat; /* medium-boundary primitive without material, scene.cpp:686-701 */
    PtrTransform to_world;
    float params[8];
    PtrMaterial mat;
    PtrMedium medium; /* interior medium (MediumInterface(medium, nullptr), scene.cpp:942) */
} PtrPrimitive;

/* ParsedMesh, src/core/shape.h:167-176.  p is in OBJECT space; the library applies
 * to_world to its own copy (the reference mutates the caller's array, mesh.cpp:55-65).
 * indices: 3 ints per corner {position, normal, uv} = 9 ints per triangle (shape.h:178-183). */
typedef struct {
    int32_t n_triangles, n_vertices, n_uvs;
    int32_t _pad;
    const float *p;         /* 3 * n_vertices */
    const int32_t *indices; /* 9 * n_triangles */
    const float *uv;        /* 2 * n_uvs or NULL */
    PtrTransform to_world;
} PtrMesh;

/* ParticleCloudDescripor, scene.h:40-44 */
typedef struct {
    int32_t n_particles;
    float radius;
    const float *positions; /* 3 * n_particles */
} PtrCloud;

/* ImageData, src/core/texture.h:10-16 */
typedef struct {
    int32_t width, height;
    const float *rgb; /* 3 * width * height, linear floats as loaded */
} PtrImage;

/* LightMap of scene.cpp:76-80 after BuildSpectrumMipMap (texture.cpp:217-260) and
 * CreateDistribution2D (infinite.cpp:24-39).  Only pyramid level 0 is reachable at
 * render time (MipMap::Lookup with width 0, texture.h:168-171). */
typedef struct {
    PtrTransform light_to_world;
    int32_t width, height; /* level-0 resolution (powers of two) */
    int32_t nu, nv;        /* distribution resolution (2*width, 2*height) */
    const float *texels;   /* 3*width*height, already multiplied by the scale */
    const float *func;     /* nv*nu              Distribution1D::func of each row      */
    const float *cdf;      /* nv*(nu+1)          Distribution1D::cdf  of each row      */
    const float *func_int; /* nv                 funcInt of each row (= marginal func) */
    const float *marg_cdf; /* nv+1                                                     */
    float marg_int;
    int32_t _pad;
} PtrEnvMap;

/* A distant light: Light::Init_Distant / Distant_Sample_Li (src/core/light.cpp:58-63, src/lights/distant.cpp:1-54) as
 * Aggregator::InsertDistantLight(L, w) queues it (src/core/primitive.cpp:180-189).  Nothing in the reference's scene API calls
 * that function; the flat scene carries such lights so that this part of src/lights is reachable.  w points FROM the
 * scene TO the light (Sample_Li returns wi = Normalize(w)). */
typedef struct {
    float L[3];
    float w[3];
} PtrDistantLight;

typedef struct {
    int32_t n_primitives, n_meshes, n_clouds, n_images;
    const PtrPrimitive *primitives; /* insertion order == hPrimitives, scene.cpp:70 */
    const PtrMesh *meshes;
    const PtrCloud *clouds;
    const PtrImage *images;
    PtrMedium camera_medium; /* is_valid = 0 when absent; InsertCameraMedium scene.cpp:721 */
    int32_t has_envmap;
    int32_t _pad;
    PtrEnvMap envmap; /* InsertEXRLightMap scene.cpp:703 */
    /* distant lights, in the light list behind the emissive primitives and the light map */
    int32_t n_distant;
    int32_t _pad2;
    const PtrDistantLight *distant;
} PtrFlatScene;

/* Camera fields, src/core/camera.h:49-57 (medium comes from the scene, main.cpp:428) */
typedef struct {
    float position[3];
    float lower_left[3];
    float horizontal[3];
    float vertical[3];
    float lens_radius;
    float u[3], v[3], w[3];
} PtrCamera;

/* Ray / hit counters: the reference's PixelStats (camera.h:12-21) summed over the
 * image as ImageStats does (camera.cpp:86-125), plus ray counts the reference does
 * not keep (every Aggregator::Intersect call equivalent). */
typedef struct {
    uint64_t total_paths;    /* extension rays = PixelStats::totalPaths, main.cpp:119,264 */
    uint64_t misses;         /* PixelStats::misses                                        */
    uint64_t surface_hits;   /* PixelStats::hits                                          */
    uint64_t medium_hits;    /* PixelStats::mediumHits                                    */
    uint64_t light_hits;     /* PixelStats::lightHits                                     */
    uint64_t zero_radiance;  /* PixelStats::zeroRadiancePaths                             */
    uint64_t shadow_rays;    /* scene queries issued by VisibilityTester (light.cpp:5-40) */
    uint64_t mis_rays;       /* scene queries issued by the BSDF-sampling half of
                                EstimateDirect (primitive.cpp:323-325, 433-461)           */
    uint64_t probe_rays;     /* BSSRDF probe segments (bssrdf.cpp:536-553)                */
    uint64_t camera_rays;    /* pixel-samples started                                     */
    uint64_t nodes_visited;  /* child boxes tested by trace kernels (profiling builds)    */
    uint64_t prims_tested;   /* leaf items tested by trace kernels (profiling builds)     */
} PtrCounters;

/* What the stages of the wavefront pipeline processed since the last ptr_counters_reset (no counterpart in the
 * reference: its megakernel has no stages).  Used by bench.py to price the kernels against their rooflines. */
typedef struct {
    uint64_t finish_paths;    /* extension rays traced inside the finish kernel (not by a tracer launch)   */
    uint64_t finish_nee_rays; /* shadow + MIS rays traced inside the finish kernel                         */
    uint64_t shaded_vertices; /* path vertices the shade kernel processed (hits that survived the miss pre-pass) */
    uint64_t nee_records;     /* of those, vertices that wrote a direct-light record                        */
    uint64_t any_nodes;       /* counting passes (ptr_profile_enable bit 1): child boxes tested by the any-hit tracer launches */
    uint64_t any_items;       /* ... leaf items / analytic primitives they tested                            */
    uint64_t any_rays;        /* ... and the rays they carried                                               */
    uint64_t fast_top_deferred; /* closest-hit rays that started in the fast top-level tree and were walked again through the
                                 * reference's tree, because their answer could depend on its order (dev_trace.h) */
} PtrPipelineStats;

/* One entry per kernel family; filled by ptr_profile_read when profiling is on. */
#define PTR_PROF_KERNELS 12
typedef struct {
    char name[32];
    uint64_t launches;
    double total_ms;
    uint64_t items; /* rays / paths processed */
} PtrKernelProfile;

typedef struct PtrContext PtrContext;

/* ---- lifetime ------------------------------------------------------------------- */

/* Replaces cudaInitEx()/cudaInit() (src/cuda/cutil.cpp:80-139): pick the device, create
 * the context.  device < 0 picks the current device. */
int ptr_create(int device, PtrContext **out);
/* Replaces ReleaseScene<<<1,1>>> + cudaFree of scene/camera (main.cpp:1150-1155). */
int ptr_destroy(PtrContext *ctx);
/* Replaces the print-and-exit error path of cutil.cpp:9-16,202-205: returns the text of
 * the last failure on this thread ("" if none). */
const char *ptr_last_error(void);
/* Use a caller-owned hipStream_t for every launch of this context (default: the null
 * stream).  No counterpart in the reference (single default stream). */
int ptr_set_stream(PtrContext *ctx, void *hip_stream);

/* ---- scene ---------------------------------------------------------------------- */

/* Replaces PrepareSceneForRendering (scene.cpp:960-987): MakeSceneGPU<<<1,1>>>
 * (scene.cpp:891), Aggregator::Wrap incl. MeshGetBounds / ParticleCloudGetBounds /
 * BuildSceneTable kernels and the three CreateBVH calls (primitive.cpp:620-642,
 * mesh.cpp:643-671, particlecloud.cpp:310-337) and MakeLights<<<1,1>>> (scene.cpp:948).
 * Flattens, builds the compact BVH on the host, uploads. */
int ptr_scene_upload(PtrContext *ctx, const PtrFlatScene *scene);
/* The same in two steps, for several contexts (GPUs) that render one scene: ptr_scene_build does the host half once
 * (flattening, both BVH builds, tables; draws from libc rand() like Aggregator::Wrap), ptr_scene_upload_built copies the
 * result to a context's device.  ptr_scene_upload = build + upload_built + free. */
typedef struct PtrBuiltScene PtrBuiltScene;
int ptr_scene_build(const PtrFlatScene *scene, PtrBuiltScene **out);
int ptr_scene_upload_built(PtrContext *ctx, const PtrBuiltScene *built);
int ptr_scene_built_free(PtrBuiltScene *built);
/* The light-map tables of PtrEnvMap built on the device: replaces BuildSpectrumMipMap (texture.cpp:111-215: scale, Lanczos
 * resample to powers of two, pyramid), the func table of texture.cpp:235-252 and CreateDistribution2D (infinite.cpp:7-39)
 * as InsertEXRLightMap runs them on the host (scene.cpp:703-719).  rgb: 3*width*height floats as read from the file,
 * rows top-down; scale3: the Spectrum the pixels are multiplied by.  The tables are the floats the host pipeline gives,
 * bit for bit (csrc/rt_envmap.hip says how).  ptr_envmap_view points a PtrEnvMap at them (light_to_world is left alone);
 * they live until ptr_envmap_free.  device < 0: the current device.  SURVEY 8f row 4. */
typedef struct PtrEnvTables PtrEnvTables;
int ptr_envmap_build(int device, const float *rgb, int width, int height, const float *scale3, PtrEnvTables **out);
int ptr_envmap_view(const PtrEnvTables *tables, PtrEnvMap *view, double *build_ms);
int ptr_envmap_free(PtrEnvTables *tables);
/* Who builds the geometry BVHs (one per mesh / particle cloud) inside ptr_scene_build / ptr_scene_upload, process-wide:
 *   PTR_BVH_HOST_SAH     binned SAH on the host (default)
 *   PTR_BVH_DEVICE_LBVH  linear BVH on the current HIP device (Morton codes, radix sort, Karras hierarchy, refit, collapse
 *                        to the same 4-wide nodes): for geometry that changes per frame (SPH clouds), where the reference's
 *                        host _CreateBVH (primitive.cpp:537-582: 14 s for 7 M triangles) is the bottleneck.  SURVEY 8f row 3.
 * Environment PTR_BVH_BUILD=device selects the second at start-up.  Traces give the same hits with either tree (hits at
 * bit-equal t aside); traversal cost differs. */
#define PTR_BVH_HOST_SAH 0
#define PTR_BVH_DEVICE_LBVH 1
int ptr_scene_build_mode(int mode);
/* builder of the uploaded scene's geometry trees, the time it took for all of them (host wall time for SAH, device time for
 * LBVH) and the depth of the deepest 4-wide tree (no counterpart; the reference prints its build time, primitive.cpp:607-615) */
typedef struct PtrBvhBuildInfo {
    int32_t builder, depth;
    double build_ms;
    /* surface-area estimate for the largest geometry tree, per ray through its root box: child boxes tested, leaf items
     * tested, leaves entered (a build-time figure of merit; bench.py measures the real per-ray counts) */
    double sah_boxes, sah_items, sah_leaves;
} PtrBvhBuildInfo;
int ptr_bvh_build_info(PtrContext *ctx, PtrBvhBuildInfo *out);
int ptr_scene_built_info(const PtrBuiltScene *built, PtrBvhBuildInfo *out); /* the same of a scene not uploaded yet */
/* 64-bit FNV-1a over the built scene's traversal tables (4-wide nodes, leaf items, top-level nodes and order): equal scenes
 * built by equal builders hash equal - the host SAH build is deterministic whatever its thread count (PTR_BUILD_THREADS).
 * No counterpart in the reference; for determinism tests and caches of built scenes. */
int ptr_scene_built_hash(const PtrBuiltScene *built, uint64_t *out);
/* Replaces Camera::Config's writes into the managed Camera (camera.cpp:145-197). */
int ptr_camera_set(PtrContext *ctx, const PtrCamera *cam);

/* ---- film ----------------------------------------------------------------------- */

/* Replaces CreateImage + SetupPixels<<<blocks,threads>>>(image, seed) (camera.cpp:134,
 * main.cpp:38-53,1110): zero accumulators, key the per-pixel RNG streams. */
int ptr_film_init(PtrContext *ctx, int width, int height, uint64_t seed);
/* Multi-GPU sharding (no counterpart; the reference is single-device): this context
 * renders only sample indices s with s % stride == offset.  Default (0,1). */
int ptr_film_shard(PtrContext *ctx, int sample_offset, int sample_stride);
/* Multi-GPU sharding by image tiles (no counterpart): the film is cut into PTR_TILE x PTR_TILE pixel tiles, numbered row
 * by row, and this context renders only the tiles t with t % tile_stride == tile_offset - every sample of their pixels,
 * with the same per-pixel RNG keys as an unsharded render.  The other pixels of its film stay as they are (zero after
 * ptr_film_init), so summing the films of all shards gives the unsharded film BIT FOR BIT (x + 0 = x), whatever the
 * number of shards.  (0, 1) = the whole film; ptr_film_init resets to that.  Needs a film.  May be combined with
 * ptr_film_shard. */
#define PTR_TILE 8
int ptr_film_tiles(PtrContext *ctx, int tile_offset, int tile_stride);
/* Replaces Render<<<blocks,threads>>>(image, scene, camera, ns) (main.cpp:421-457,1139):
 * ns more samples for every pixel with integrator PTR_LI_*.  Asynchronous, and ordered with the
 * context's stream (ptr_set_stream; default: the null stream) on both sides: the passes run on internal
 * streams; they read and write the film (their last kernel) only after everything queued on the context's stream
 * before this call, and work queued on the context's stream after this call waits for the passes. */
int ptr_render(PtrContext *ctx, int ns, int integrator);
/* Replaces cudaDeviceAssert() after a launch (main.cpp:1140). */
int ptr_sync(PtrContext *ctx);
/* Replaces the host's reads of the managed Pixel array (GetPixelRGB camera.cpp:36-51,
 * graphy.cpp:71-93): copies Pixel::we (3 floats/pixel), Pixel::accWeight, Pixel::samples
 * to host arrays (any may be NULL).  Synchronises. */
int ptr_film_read(PtrContext *ctx, float *we_rgb, double *acc_weight, int64_t *samples);
/* Replaces the loop of ImageWrite over GetPixelRGB (camera.cpp:36-51, 53-68): resolves the film on
 * the device - e = we / accWeight, tone map PTR_TONEMAP_* with `exposure` (ReinhardMap /
 * ExponentialMap / NaughtyDogMap, camera.cpp:6-33), sRGB GammaCorrect (geometry.h:73-76),
 * (int)(255.999 * e) narrowed to a byte - and copies width*height*3 bytes to `rgb` in pixel order
 * (row 0 = bottom of the frame, like the Pixel array).  Pixels with zero accWeight give 0 (the
 * reference asserts).  Synchronises.  SURVEY section 8(f) row 2. */
#define PTR_TONEMAP_REINHARD 0
#define PTR_TONEMAP_EXPONENTIAL 1
#define PTR_TONEMAP_NAUGHTYDOG 2
int ptr_film_rgb8(PtrContext *ctx, float exposure, int algorithm, unsigned char *rgb);
/* Inverse of ptr_film_read: overwrites the accumulators of every pixel (host arrays of the same
 * shapes).  The reference keeps Pixel[] in managed memory, so host code that restores a saved film
 * writes image->pixels directly (camera.h:23-36); this is that write.  A film restored this way and
 * rendered further continues the sample sequence: the next ptr_render call starts at sample index
 * `next_sample` (-1: take samples[0], right for a film that holds every sample so far; ranks of a
 * sharded render that restart from an empty film pass the saved index).  Synchronises. */
int ptr_film_write(PtrContext *ctx, const float *we_rgb, const double *acc_weight, const int64_t *samples, int64_t next_sample);
/* Device addresses of the accumulators, for an RCCL reduce across ranks (no counterpart). */
int ptr_film_device_ptrs(PtrContext *ctx, void **we_rgb, void **acc_weight, void **samples);
/* Gathering the film of a tile-split render without arithmetic (no counterpart): a shard's own tiles, packed.
 * The packed film of a shard holds its tiles t = offset + k * stride back to back, 64 pixels per tile (pixels of an edge tile
 * that lie outside the frame are zero), pixel (x, y) of tile k at entry k * 64 + (y & 7) * 8 + (x & 7): we[3 * entry],
 * acc[entry], samples[entry].  Every shard of a `stride`-way split packs ptr_film_tiles_packed_pixels(stride) entries, so
 * the packed films are one fixed-size gather (ncclSend / ncclRecv, all_gather) instead of an all-reduce of whole films:
 * 1 / stride of the bytes per rank, the rank's film stays as it is, and the gathered film can be assembled any number
 * of times.  ptr_film_tiles_pack packs THIS context's film for its current ptr_film_tiles shard; ptr_film_tiles_unpack
 * writes the pixels of shard (tile_offset, tile_stride) from a packed film into full-frame accumulators d_out_* of the
 * context's width x height (any device buffers, e.g. a result film kept apart from every rank's own).  Both are
 * asynchronous on the context's stream and allocate nothing. */
int ptr_film_tiles_packed_pixels(PtrContext *ctx, int tile_stride, int64_t *out);
int ptr_film_tiles_pack(PtrContext *ctx, float *d_we, double *d_acc, int64_t *d_samples);
int ptr_film_tiles_unpack(PtrContext *ctx, int tile_offset, int tile_stride, const float *d_we, const double *d_acc, const int64_t *d_samples,
                          float *d_out_we, double *d_out_acc, int64_t *d_out_samples);
/* d_out_samples[0 .. width*height) = value: the sample count of a sample-split render is known without a reduce
 * (no counterpart).  Asynchronous on the context's stream. */
int ptr_film_fill_samples(PtrContext *ctx, int64_t *d_out_samples, int64_t value);
/* Replaces ImageStats (camera.cpp:86-125). Synchronises. */
int ptr_counters_read(PtrContext *ctx, PtrCounters *out);
int ptr_counters_reset(PtrContext *ctx);
/* Stage statistics (above); cleared by ptr_counters_reset.  Synchronises. */
int ptr_pipeline_stats_read(PtrContext *ctx, PtrPipelineStats *out);

/* ---- hot-path stages exposed for parity tests and benchmarks ------------------------ */

/* Ray generation half of Render (main.cpp:436-443 + Camera::SpawnRay camera.cpp:219-224)
 * for sample index `sample` of every pixel.  Device pointers, n = width*height entries:
 * o[3n] d[3n] as SoA planes (x plane, y plane, z plane), weight[n] = GetFilterWeight. */
int ptr_generate_rays(PtrContext *ctx, int sample, float *d_o, float *d_d, float *d_weight);
/* Aggregator::Intersect (primitive.cpp:374-431) over a batch of n rays resident in HBM.
 * SoA planes: d_o[3n], d_d[3n], d_tmax[n].  Outputs (device): prim[n] (-1 = miss),
 * face[n] (triangle / particle / sdf shape id), t[n], b0[n], b1[n]. */
int ptr_trace_closest(PtrContext *ctx, int n, const float *d_o, const float *d_d,
                      const float *d_tmax, int32_t *d_prim, int32_t *d_face, float *d_t,
                      float *d_b0, float *d_b1);
/* ptr_trace_closest for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments, same results bit for
 * bit; it calls no hipMalloc / hipFree (nor any allocating library routine) and does not synchronise: one 256-byte
 * memset and one kernel launch on the context's stream.  The tracer's ray-fetch cursor - the only device memory the
 * call needs beyond its inputs and outputs - lives in a 256-byte `static __device__` array of the library (one copy per
 * GPU, held until the process ends) instead of the hipMalloc'ed ring ptr_trace_closest uses.  Limits: at most
 * PTR_TRACE_NOALLOC_MAX_RAYS rays per call (the cursor is a 32-bit int and runs past n by one grab per resident
 * wavefront; more rays: PTR_ERR_LIMIT, nothing is launched); every call on a GPU shares that GPU's array, so two
 * calls that overlap on one GPU (different contexts or streams) race - run them one after the other on one stream. */
#define PTR_TRACE_NOALLOC_MAX_RAYS (1 << 30)
int ptr_trace_closest_noalloc(PtrContext *ctx, int n, const float *d_o, const float *d_d,
                              const float *d_tmax, int32_t *d_prim, int32_t *d_face, float *d_t,
                              float *d_b0, float *d_b1);
/* VisibilityTester::Unoccluded (light.cpp:5-8): d_occluded[n] = 1 if anything is hit. */
int ptr_trace_occluded(PtrContext *ctx, int n, const float *d_o, const float *d_d,
                       const float *d_tmax, int32_t *d_occluded);

/* ---- unit hooks for the known-answer tests (no counterpart: the reference has no tests) ------------
 * The building blocks of the shade stage on plain device arrays of n records (`in_width` floats in, `out_width` floats
 * out per record), so that tests can integrate, histogram and compare them with closed forms:
 *   kind                      in                                out          reference
 *   PTR_UNIT_FR_DIELECTRIC    cosI etaI etaT                    F            material.cpp:5-30
 *   PTR_UNIT_FR_CONDUCTOR     cosI etaI(3) etaT(3) k(3)         F(3)         material.cpp:32-52
 *   PTR_UNIT_GGX_D            ax ay wh(3)                       D            microfacet.cpp:3-21
 *   PTR_UNIT_GGX_G1           ax ay w(3)                        G1           microfacet.h:30-36
 *   PTR_UNIT_GGX_PDF          ax ay wo(3) wh(3)                 pdf          microfacet.cpp (Pdf, visible normals)
 *   PTR_UNIT_GGX_SAMPLE_WH    ax ay wo(3) u(2)                  wh(3)        microfacet.cpp:23-120
 *   PTR_UNIT_HG_P             cosTheta g                        p            medium.cpp:4-19
 *   PTR_UNIT_HG_SAMPLE        g wo(3) u(2)                      wi(3) p      medium.cpp:21-64
 *   PTR_UNIT_ROUGH_TO_ALPHA   roughness                         alpha        microfacet.h:38-43
 *   PTR_UNIT_ENV_SAMPLE       u(2)                              uv(2) pdf    sampling.h:49-77 (the scene's light map)
 *   PTR_UNIT_ENV_PDF          uv(2)                             pdf          sampling.h:63-69
 *   PTR_UNIT_SSS_SR           r                                 Sr(3)        bssrdf.cpp:383-411 (material of `prim`)
 *   PTR_UNIT_SSS_PDF_SR       ch r                              pdf          bssrdf.cpp:413-437
 *   PTR_UNIT_SSS_SAMPLE_SR    ch u                              r            bssrdf.cpp:439-445
 * Asynchronous on the context's stream. */
enum {
    PTR_UNIT_FR_DIELECTRIC = 0, PTR_UNIT_FR_CONDUCTOR, PTR_UNIT_GGX_D, PTR_UNIT_GGX_G1, PTR_UNIT_GGX_PDF, PTR_UNIT_GGX_SAMPLE_WH,
    PTR_UNIT_HG_P, PTR_UNIT_HG_SAMPLE, PTR_UNIT_ROUGH_TO_ALPHA, PTR_UNIT_ENV_SAMPLE, PTR_UNIT_ENV_PDF, PTR_UNIT_SSS_SR,
    PTR_UNIT_SSS_PDF_SR, PTR_UNIT_SSS_SAMPLE_SR, PTR_UNIT_KINDS
};
int ptr_unit_eval(PtrContext *ctx, int prim, int kind, int n, const float *d_in, int in_width, float *d_out, int out_width);
/* BSDF::f and BSDF::Pdf over the non-specular lobes (sample = 0: d_wi_or_u holds n directions wi; d_wi_out / d_flags unused)
 * or BSDF::Sample_f over all lobes (sample = 1: d_wi_or_u holds n pairs u; wi, f, pdf and the sampled lobe's type flags
 * come back) of the material of primitive `prim` (ComputeScatteringFunctions, material.cpp:103-165 ->
 * reflection.cpp:591-701) at a canonical interaction: p = 0, n = +z, dpdu = +x, uv from d_uv[2n].  World-space
 * directions wo[3n] (AoS).  No counterpart in the reference.  Asynchronous on the context's stream. */
int ptr_unit_bsdf(PtrContext *ctx, int prim, int sample, int n, const float *d_wo, const float *d_wi_or_u, const float *d_uv,
                  float *d_wi_out, float *d_f, float *d_pdf, int32_t *d_flags);

/* ---- introspection (no counterpart) ----------------------------------------------- */
int ptr_profile_enable(PtrContext *ctx, int on);
int ptr_profile_reset(PtrContext *ctx);
int ptr_profile_read(PtrContext *ctx, PtrKernelProfile *out, int max_entries, int *n_out);
/* nodes, leaf items, bytes of the shipped BVH */
int ptr_bvh_info(PtrContext *ctx, int64_t *n_nodes, int64_t *n_items, int64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif /* PTR_API_H */
