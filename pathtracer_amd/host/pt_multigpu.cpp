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

/* One packed film (ptr_film_tiles_pack): [we 3 x f32 | accWeight f64 | samples i64] x entries, one allocation so that it
 * travels as one message */
struct PackedFilm {
    char *base = nullptr;
    size_t entries = 0;
    size_t bytes() const { return entries * 28; }
    float *we() const { return reinterpret_cast<float *>(base); }
    double *acc() const { return reinterpret_cast<double *>(base + entries * 12); }
    int64_t *smp() const { return reinterpret_cast<int64_t *>(base + entries * 20); }
};
struct RenderGroup {
    int n = 0, split = PT_SPLIT_TILES;
    std::vector<int> devices;
    std::vector<PtrContext *> ctx;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;
    int width = 0, height = 0;
    int64_t samples_done = 0; /* samples of every pixel rendered by the group since RenderGroupFilmInit */
    /* the reduced film lives on GPU 0, apart from every GPU's own (partial) film: reducing never changes what a GPU
     * has accumulated, so render -> reduce -> render -> reduce shows the film after every step (main.cpp:1135-1146) */
    float *res_we = nullptr;
    double *res_acc = nullptr;
    int64_t *res_smp = nullptr;
    std::vector<PackedFilm> packed; /* tile split: per GPU, its own tiles */
    std::vector<PackedFilm> inbox;  /* tile split: on GPU 0, where the packed films of GPUs 1 .. n-1 arrive */
    bool reduced = false;
};
static void free_result(RenderGroup *g) {
    if (g->n > 0) (void)hipSetDevice(g->devices[0]);
    for (void *p : {(void *)g->res_we, (void *)g->res_acc, (void *)g->res_smp})
        if (p) (void)hipFree(p);
    g->res_we = nullptr; g->res_acc = nullptr; g->res_smp = nullptr;
    for (PackedFilm &f : g->inbox)
        if (f.base) (void)hipFree(f.base);
    g->inbox.clear();
    for (size_t i = 0; i < g->packed.size(); i++)
        if (g->packed[i].base) {
            (void)hipSetDevice(g->devices[i]);
            (void)hipFree(g->packed[i].base);
        }
    g->packed.clear();
    g->reduced = false;
}

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
    free_result(g);
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
    if (width != g->width || height != g->height || !g->res_we) {
        free_result(g);
        const size_t npix = (size_t)width * height;
        GHIP(hipSetDevice(g->devices[0]));
        GHIP(hipMalloc((void **)&g->res_we, 3 * sizeof(float) * npix));
        GHIP(hipMalloc((void **)&g->res_acc, sizeof(double) * npix));
        GHIP(hipMalloc((void **)&g->res_smp, sizeof(int64_t) * npix));
        if (g->split == PT_SPLIT_TILES) {
            int64_t entries = 0;
            GPTR(ptr_film_tiles_packed_pixels(g->ctx[0], g->n, &entries));
            g->packed.assign(g->n, PackedFilm());
            g->inbox.assign(g->n, PackedFilm());
            for (int i = 0; i < g->n; i++) {
                g->packed[i].entries = (size_t)entries;
                GHIP(hipSetDevice(g->devices[i]));
                GHIP(hipMalloc((void **)&g->packed[i].base, g->packed[i].bytes()));
                if (i > 0) {
                    g->inbox[i].entries = (size_t)entries;
                    GHIP(hipSetDevice(g->devices[0]));
                    GHIP(hipMalloc((void **)&g->inbox[i].base, g->inbox[i].bytes()));
                }
            }
        }
    }
    g->width = width;
    g->height = height;
    g->samples_done = 0;
    g->reduced = false;
    return PTR_OK;
}
int RenderGroupRenderSamples(RenderGroup *g, int ns, int integrator) {
    if (!g) return gfail(PTR_ERR_INVALID, "null group");
    for (int i = 0; i < g->n; i++) GPTR(ptr_render(g->ctx[i], ns, integrator)); /* asynchronous: all GPUs work at once */
    g->samples_done += ns;
    return PTR_OK;
}
/* The film of the whole group, assembled on GPU 0 into a buffer of its own.  Asynchronous: ordered behind every GPU's passes
 * (ptr_render joined them into that GPU's stream).  May be called any number of times, between renders too.
 *   tiles:    every GPU packs its own tiles (1 / n of the film); GPUs 1 .. n-1 send theirs to GPU 0 (one grouped
 *             ncclSend / ncclRecv exchange over xGMI, 28 B per pixel in total), GPU 0 scatters them into the result: no
 *             arithmetic, so the result is the one-GPU film bit for bit;
 *   samples:  ncclReduce(sum) of we and accWeight to GPU 0; the sample count is known (samples_done), not reduced. */
int RenderGroupReduce(RenderGroup *g) {
    if (!g) return gfail(PTR_ERR_INVALID, "null group");
    if (!g->res_we) return gfail(PTR_ERR_STATE, "RenderGroupReduce needs RenderGroupFilmInit");
    const size_t npix = (size_t)g->width * g->height;
    if (g->split == PT_SPLIT_TILES) {
        for (int i = 0; i < g->n; i++) GPTR(ptr_film_tiles_pack(g->ctx[i], g->packed[i].we(), g->packed[i].acc(), g->packed[i].smp()));
        if (g->n > 1) {
            GNCCL(ncclGroupStart());
            for (int i = 1; i < g->n; i++) {
                GNCCL(ncclSend(g->packed[i].base, g->packed[i].bytes(), ncclChar, 0, g->comms[i], g->streams[i]));
                GNCCL(ncclRecv(g->inbox[i].base, g->inbox[i].bytes(), ncclChar, i, g->comms[0], g->streams[0]));
            }
            GNCCL(ncclGroupEnd());
        }
        for (int i = 0; i < g->n; i++) {
            const PackedFilm &f = i == 0 ? g->packed[0] : g->inbox[i];
            GPTR(ptr_film_tiles_unpack(g->ctx[0], i, g->n, f.we(), f.acc(), f.smp(), g->res_we, g->res_acc, g->res_smp));
        }
    } else {
        std::vector<void *> we(g->n), acc(g->n);
        for (int i = 0; i < g->n; i++) GPTR(ptr_film_device_ptrs(g->ctx[i], &we[i], &acc[i], nullptr));
        GNCCL(ncclGroupStart());
        for (int i = 0; i < g->n; i++) {
            GNCCL(ncclReduce(we[i], i == 0 ? g->res_we : nullptr, 3 * npix, ncclFloat, ncclSum, 0, g->comms[i], g->streams[i]));
            GNCCL(ncclReduce(acc[i], i == 0 ? g->res_acc : nullptr, npix, ncclDouble, ncclSum, 0, g->comms[i], g->streams[i]));
        }
        GNCCL(ncclGroupEnd());
        GPTR(ptr_film_fill_samples(g->ctx[0], g->res_smp, g->samples_done));
    }
    g->reduced = true;
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
    if (!g->reduced) return gfail(PTR_ERR_STATE, "RenderGroupFilmRead needs RenderGroupReduce after the last render");
    int rc = RenderGroupSync(g);
    if (rc) return rc;
    const size_t npix = (size_t)g->width * g->height;
    GHIP(hipSetDevice(g->devices[0]));
    if (we_rgb) GHIP(hipMemcpy(we_rgb, g->res_we, 3 * sizeof(float) * npix, hipMemcpyDeviceToHost));
    if (acc_weight) GHIP(hipMemcpy(acc_weight, g->res_acc, sizeof(double) * npix, hipMemcpyDeviceToHost));
    if (samples) GHIP(hipMemcpy(samples, g->res_smp, sizeof(int64_t) * npix, hipMemcpyDeviceToHost));
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
    /* the per-GPU films are partial; the reduce assembles them into the group's result film and leaves them as they are, so
     * this may follow every Render call, as the reference's loop shows the picture after every sample (main.cpp:1135-1146) */
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
