/*
 * pt_multigpu.h — one scene rendered by G GPUs of one node from a single C++ host process.
 *
 * The reference is single-device (cudaChooseDevice, src/cuda/cutil.cpp:84-139); nothing here has a counterpart in it.
 * The scene is built once on the host and replicated on every GPU (ptr_scene_build / ptr_scene_upload_built); the work
 * is split without any exchange while rendering, either
 *   PT_SPLIT_TILES    8x8-pixel tiles dealt round-robin (tile t -> GPU t mod G), every sample of a tile on one GPU
 *                     with the per-pixel RNG keys of a 1-GPU render: the reduced film is bit-identical to it for any G;
 *   PT_SPLIT_SAMPLES  sample s -> GPU s mod G (better balance on very uneven images; sums differ in rounding);
 * and the float accumulators (Pixel::we, accWeight, samples) are assembled on GPU 0 in a result film of their own - tiles:
 * every GPU packs its tiles and one grouped ncclSend / ncclRecv exchange gathers them (1 / G of the film per GPU, no
 * arithmetic); samples: ncclReduce(sum) of we and accWeight, the sample count being known - on streams the renders are
 * already joined into, so no host sync is needed between rendering and reducing.  A GPU's own film is never written by
 * the reduce: it can be repeated, and rendering can go on after it (the reference shows the film after every sample,
 * main.cpp:1135-1146).
 */
#pragma once
#include "../../include/ptr_api.h"
#include <cstdint>

enum { PT_SPLIT_TILES = 0, PT_SPLIT_SAMPLES = 1 };
struct RenderGroup;

#ifdef __cplusplus
extern "C" {
#endif
/* devices = null: GPUs 0 .. n_gpus-1.  Returns null (and prints why) on failure. */
RenderGroup *RenderGroupCreate(int n_gpus, const int *devices, int split);
void RenderGroupFree(RenderGroup *g);
int RenderGroupSize(const RenderGroup *g);
PtrContext *RenderGroupContext(RenderGroup *g, int i);
/* flat-scene level (what the C++ mirror below and the Python tests both use); all return PTR_OK or a PTR_ERR_* code */
int RenderGroupUpload(RenderGroup *g, const PtrFlatScene *scene);
int RenderGroupCamera(RenderGroup *g, const PtrCamera *cam);
int RenderGroupFilmInit(RenderGroup *g, int width, int height, uint64_t seed);
int RenderGroupRenderSamples(RenderGroup *g, int ns, int integrator); /* asynchronous on every GPU */
int RenderGroupReduce(RenderGroup *g);                                 /* the group's film assembled on GPU 0 (gather of tiles / reduce of samples), asynchronous, repeatable */
int RenderGroupSync(RenderGroup *g);
int RenderGroupFilmRead(RenderGroup *g, float *we_rgb, double *acc_weight, int64_t *samples); /* the result film of the last Reduce */
const char *RenderGroupLastError(void);
#ifdef __cplusplus
}
/* the reference's entry points (scene.h:88-185, main.cpp:38-53, 421-457) over a group: same argument meaning as
 * PrepareSceneForRendering / SetupPixels / Render / ImageSync in pt_scene.h */
struct Image;
struct Camera;
void PrepareSceneForRendering(RenderGroup *g);
void SetupPixels(RenderGroup *g, Image *image, unsigned long long seed);
void Render(RenderGroup *g, Image *image, Camera *camera, int ns); /* renders (asynchronous); ImageSync reduces and reads back */
void ImageSync(RenderGroup *g, Image *image);
#endif
