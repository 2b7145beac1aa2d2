/*
 * pt_multigpu.cpp — see pt_multigpu.h.  RCCL is called directly (no Python, no torch.distributed): one communicator
 * per GPU made with ncclCommInitAll, one host thread issuing the grouped collectives for all of them.
 */
#define __HIP_PLATFORM_AMD__ 1
#include "pt_multigpu.h"
#include "pt_scene.h"
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

struct RenderGroup {
    int n = 0, split = PT_SPLIT_TILES;
    std::vector<int> devices;
    std::vector<PtrContext *> ctx;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;
    int width = 0, height = 0;
};

static thread_local std::string g_gerr;
static int gfail(int code, const std::string &msg) {
    g_gerr = msg;
    return code;
}
#define GPTR(expr)                                                                          \
    do {                                                                                    \
        int rc_ = (expr);                                                                   \
        if (rc_ != PTR_OK) return gfail(rc_, std::string(#expr) + ": " + ptr_last_error()); \
    } while (0)
#define GHIP(expr)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return gfail(PTR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define GNCCL(expr)                                                                                     \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) return gfail(PTR_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)

extern "C" {

const char *RenderGroupLastError(void) { return g_gerr.c_str(); }

RenderGroup *RenderGroupCreate(int n_gpus, const int *devices, int split) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || n_gpus < 1 || n_gpus > ndev || (split != PT_SPLIT_TILES && split != PT_SPLIT_SAMPLES)) {
        fprintf(stderr, "RenderGroupCreate: %d GPU(s) asked for, %d visible, split %d\n", n_gpus, ndev, split);
        return nullptr;
    }
    RenderGroup *g = new RenderGroup();
    g->n = n_gpus;
    g->split = split;
    for (int i = 0; i < n_gpus; i++) g->devices.push_back(devices ? devices[i] : i);
    g->ctx.assign(n_gpus, nullptr);
    g->streams.assign(n_gpus, nullptr);
    g->comms.assign(n_gpus, nullptr);
    bool ok = true;
    for (int i = 0; i < n_gpus && ok; i++) {
        ok = ptr_create(g->devices[i], &g->ctx[i]) == PTR_OK && hipSetDevice(g->devices[i]) == hipSuccess &&
             hipStreamCreateWithFlags(&g->streams[i], hipStreamNonBlocking) == hipSuccess && ptr_set_stream(g->ctx[i], g->streams[i]) == PTR_OK;
    }
    if (ok) ok = ncclCommInitAll(g->comms.data(), n_gpus, g->devices.data()) == ncclSuccess;
    if (!ok) {
        fprintf(stderr, "RenderGroupCreate failed: %s\n", ptr_last_error());
        RenderGroupFree(g);
        return nullptr;
    }
    return g;
}

void RenderGroupFree(RenderGroup *g) {
    if (!g) return;
    for (int i = 0; i < g->n; i++) {
        if (g->comms[i]) ncclCommDestroy(g->comms[i]);
        if (g->ctx[i]) ptr_destroy(g->ctx[i]);
        if (g->streams[i]) {
            (void)hipSetDevice(g->devices[i]);
            (void)hipStreamDestroy(g->streams[i]);
        }
    }
    delete g;
}
int RenderGroupSize(const RenderGroup *g) { return g ? g->n : 0; }
PtrContext *RenderGroupContext(RenderGroup *g, int i) { return g && i >= 0 && i < g->n ? g->ctx[i] : nullptr; }

int RenderGroupUpload(RenderGroup *g, const PtrFlatScene *scene) {
    if (!g) return gfail(PTR_ERR_INVALID, "null group");
    PtrBuiltScene *built = nullptr;
    GPTR(ptr_scene_build(scene, &built)); /* host half once: flatten, BVH builds, tables */
    int rc = PTR_OK;
    for (int i = 0; i < g->n && rc == PTR_OK; i++) rc = ptr_scene_upload_built(g->ctx[i], built);
    ptr_scene_built_free(built);
    if (rc != PTR_OK) return gfail(rc, std::string("ptr_scene_upload_built: ") + ptr_last_error());
    return PTR_OK;
}
int RenderGroupCamera(RenderGroup *g, const PtrCamera *cam) {
    if (!g) return gfail(PTR_ERR_INVALID, "null group");
    for (int i = 0; i < g->n; i++) GPTR(ptr_camera_set(g->ctx[i], cam));
    return PTR_OK;
}
int RenderGroupFilmInit(RenderGroup *g, int width, int height, uint64_t seed) {
    if (!g) return gfail(PTR_ERR_INVALID, "null group");
    for (int i = 0; i < g->n; i++) {
        GPTR(ptr_film_init(g->ctx[i], width, height, seed));
        if (g->split == PT_SPLIT_TILES) GPTR(ptr_film_tiles(g->ctx[i], i, g->n));
        else GPTR(ptr_film_shard(g->ctx[i], i, g->n));
    }
    g->width = width;
    g->height = height;
    return PTR_OK;
}
int RenderGroupRenderSamples(RenderGroup *g, int ns, int integrator) {
    if (!g) return gfail(PTR_ERR_INVALID, "null group");
    for (int i = 0; i < g->n; i++) GPTR(ptr_render(g->ctx[i], ns, integrator)); /* asynchronous: all GPUs work at once */
    return PTR_OK;
}
int RenderGroupReduce(RenderGroup *g) {
    if (!g) return gfail(PTR_ERR_INVALID, "null group");
    const size_t npix = (size_t)g->width * g->height;
    std::vector<void *> we(g->n), acc(g->n), smp(g->n);
    for (int i = 0; i < g->n; i++) GPTR(ptr_film_device_ptrs(g->ctx[i], &we[i], &acc[i], &smp[i]));
    /* one group: the three reductions of every GPU are fused into one launch per GPU, ordered behind that GPU's passes
     * (ptr_render joined them into its stream) */
    GNCCL(ncclGroupStart());
    for (int i = 0; i < g->n; i++) {
        GNCCL(ncclAllReduce(we[i], we[i], 3 * npix, ncclFloat, ncclSum, g->comms[i], g->streams[i]));
        GNCCL(ncclAllReduce(acc[i], acc[i], npix, ncclDouble, ncclSum, g->comms[i], g->streams[i]));
        GNCCL(ncclAllReduce(smp[i], smp[i], npix, ncclInt64, ncclSum, g->comms[i], g->streams[i]));
    }
    GNCCL(ncclGroupEnd());
    return PTR_OK;
}
int RenderGroupSync(RenderGroup *g) {
    if (!g) return gfail(PTR_ERR_INVALID, "null group");
    for (int i = 0; i < g->n; i++) {
        GPTR(ptr_sync(g->ctx[i]));
        GHIP(hipSetDevice(g->devices[i]));
        GHIP(hipStreamSynchronize(g->streams[i]));
    }
    return PTR_OK;
}
int RenderGroupFilmRead(RenderGroup *g, float *we_rgb, double *acc_weight, int64_t *samples) {
    if (!g) return gfail(PTR_ERR_INVALID, "null group");
    int rc = RenderGroupSync(g);
    if (rc) return rc;
    GPTR(ptr_film_read(g->ctx[0], we_rgb, acc_weight, samples));
    return PTR_OK;
}

} /* extern "C" */

/* ---- the reference's entry points over a group ---- */
static int g_group_integrator = PTR_LI_VOLPATH;
static void gdie(int rc, const char *what) {
    if (rc == PTR_OK) return;
    fprintf(stderr, "%s failed: %s\n", what, RenderGroupLastError());
    exit(1); /* the reference's error path: print and exit (cutil.cpp:9-16) */
}
void PrepareSceneForRendering(RenderGroup *g) {
    FlatSceneHolder *h = FlattenScene();
    gdie(RenderGroupUpload(g, FlatSceneGet(h)), "RenderGroupUpload");
    FlatSceneFree(h);
}
void SetupPixels(RenderGroup *g, Image *image, unsigned long long seed) {
    gdie(RenderGroupFilmInit(g, image->width, image->height, seed), "RenderGroupFilmInit");
}
void Render(RenderGroup *g, Image *image, Camera *camera, int ns) {
    (void)image;
    PtrCamera pc = CameraToFlat(*camera);
    gdie(RenderGroupCamera(g, &pc), "RenderGroupCamera");
    gdie(RenderGroupRenderSamples(g, ns, g_group_integrator), "RenderGroupRenderSamples");
}
void ImageSync(RenderGroup *g, Image *image) {
    /* the per-GPU films are partial until reduced; the reduce adds the films, so it is done once, here, on the way out */
    gdie(RenderGroupReduce(g), "RenderGroupReduce");
    size_t n = (size_t)image->pixels_count;
    std::vector<float> we(3 * n);
    std::vector<double> aw(n);
    std::vector<int64_t> sm(n);
    gdie(RenderGroupFilmRead(g, we.data(), aw.data(), sm.data()), "RenderGroupFilmRead");
    for (size_t i = 0; i < n; i++) {
        image->pixels[i].we = Spectrum(we[3 * i], we[3 * i + 1], we[3 * i + 2]);
        image->pixels[i].accWeight = aw[i];
        image->pixels[i].samples = (long)sm[i];
    }
}
