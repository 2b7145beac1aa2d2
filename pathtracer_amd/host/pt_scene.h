/*
 * pt_scene.h — the reference's scene-description and render entry-point API, kept verbatim
 * (names, argument order, defaults) so scene functions written against src/core/scene.h:88-185,
 * src/core/camera.h:38-80 and src/main.cpp's render() drop in unchanged.  What differs is behind
 * the API: descriptors are flattened on the host and handed to the HIP library through the
 * C-ABI of include/ptr_api.h instead of being `new`-ed inside a <<<1,1>>> kernel.
 */
#pragma once
#include "../../include/ptr_api.h"
#include "pt_math.h"
#include <string>
#include <vector>

enum ShapeType { /* shape.h:8-11 */
    SPHERE, RECTANGLE, DISK, BOX, MESH, PARTICLECLOUD, SPHERE_PROCEDURAL, BOX_PROCEDURAL, COMPONENT_PROCEDURAL,
};
enum MaterialType { Matte = 1, Mirror, Glass, Metal, Translucent, Plastic, Uber, Subsurface, KdSubsurface, PlayGround, Disney };
enum ToneMapAlgorithm { Reinhard, Exponential, NaughtyDog }; /* camera.h:6-10 */

struct ImageData { /* texture.h:10-16 */
    Spectrum *data;
    int width, height;
    int is_valid;
    Float GetAspectRatio() { return (Float)(width) / (Float)(height); }
};

typedef struct { /* shape.h:167-176 */
    Point3f *p;
    Normal3f *n;
    vec3f *s;
    Point2f *uv;
    Point3i *indices;
    int nTriangles, nVertices;
    int nUvs, nNormals;
    Transform toWorld;
} ParsedMesh;

/* scene.h:11-86 */
typedef struct { Transform toWorld; Float radius; bool reverseOrientation; bool procedural; } SphereDescriptor;
typedef struct { ParsedMesh *mesh; } MeshDescriptor;
typedef struct { Float sizex, sizey; Transform toWorld; bool reverseOrientation; } RectDescriptor;
typedef struct { Transform toWorld; Float sizex, sizey, sizez; bool reverseOrientation; } BoxDescriptor;
typedef struct { Transform toWorld; Float height, radius, innerRadius, phiMax; bool reverseOrientation; } DiskDescriptor;
typedef struct { vec3f *pos; int npos; Float scale; } ParticleCloudDescripor;
typedef struct { Transform toWorld; int id; Bounds3f bound; } ProcToyDescriptor;
typedef struct { Spectrum spec_val; Float fval; ImageData *image; int type; } TextureDescriptor;
typedef struct {
    int is_emissive;
    int specs_taken;
    MaterialType type;
    TextureDescriptor textures[16];
    Float vals[16];
    int id;
} MaterialDescriptor;
typedef struct { Spectrum sa, ss; Float g; int is_valid; } MediumDescriptor;
typedef struct {
    ShapeType shapeType;
    SphereDescriptor sphereDesc;
    MeshDescriptor meshDesc;
    RectDescriptor rectDesc;
    BoxDescriptor boxDesc;
    DiskDescriptor diskDesc;
    ProcToyDescriptor toyDesc;
    ParticleCloudDescripor partDesc;
    MaterialDescriptor mat;
    MediumDescriptor mediumDesc;
    int no_mat;
} PrimitiveDescriptor;

/* The reference's Aggregator is a device object graph (primitive.h:77-130); here it is the
 * host handle of one GPU context.  Image keeps the reference's fields that host code reads. */
struct Aggregator {
    PtrContext *ctx;
    int device;
    Aggregator() : ctx(nullptr), device(-1) {}
};
struct Pixel { /* camera.h:23-30, host mirror filled by ImageSync */
    Spectrum we;
    double accWeight;
    long samples;
};
struct Image { /* camera.h:32-36 */
    Pixel *pixels;
    int pixels_count;
    int width, height;
    Aggregator *owner; /* context whose film this image mirrors (set by SetupPixels) */
};

class Camera { /* camera.h:49-80 */
  public:
    Point3f position;
    Point3f lower_left;
    vec3f horizontal;
    vec3f vertical;
    Float lensRadius;
    vec3f u, v, w;
    Camera() : lensRadius(0) {}
    void Config(Point3f eye, Point3f at, vec3f up, Float fov, Float aspect);
    void Config(Point3f eye, Point3f at, vec3f up, Float fov, Float aspect, Float aperture);
    void Config(Point3f eye, Point3f at, vec3f up, Float fov, Float aspect, Float aperture, Float focus_dist);
};

/* ---- scene.h:88-185 ---------------------------------------------------------------------- */
void BeginScene(Aggregator *scene);
SphereDescriptor MakeSphere(Transform toWorld, Float radius, bool reverseOrientation = false);
SphereDescriptor MakeSphereProcedural(Transform toWorld, Float radius);
RectDescriptor MakeRectangle(Transform toWorld, Float sizex, Float sizey, bool reverseOrientation = false);
MeshDescriptor MakeMesh(ParsedMesh *mesh);
BoxDescriptor MakeBox(Transform toWorld, Float sizex, Float sizey, Float sizez, bool reverseOrientation = false);
DiskDescriptor MakeDisk(Transform toWorld, Float height, Float radius, Float innerRadius, Float phiMax,
                        bool reverseOrientation = false);
ProcToyDescriptor MakeProceduralToy(Transform toWorld, Bounds3f bound, int id);
ParticleCloudDescripor MakeParticleCloud(vec3f *positions, int n, Float size);

TextureDescriptor MakeTexture(Spectrum value);
TextureDescriptor MakeTexture(Float value);
TextureDescriptor MakeTexture(ImageData *data);

MaterialDescriptor MakeMatteMaterial(Spectrum kd, Float sigma = 0);
MaterialDescriptor MakeMatteMaterial(TextureDescriptor kd, Float sigma = 0);
MaterialDescriptor MakeMirrorMaterial(Spectrum kr);
MaterialDescriptor MakeMetalMaterial(Spectrum kr, Spectrum k, Float etaI, Float etaT);
MaterialDescriptor MakeGlassMaterial(Spectrum kr, Spectrum kt, Float index, Float uRough = 0, Float vRough = 0);
MaterialDescriptor MakeGlassMaterial(Spectrum kr, Spectrum kt, Float index, TextureDescriptor rough);
MaterialDescriptor MakeGlassMaterial(TextureDescriptor kr, TextureDescriptor kt, Float index, TextureDescriptor rough);
MaterialDescriptor MakeSubsurfaceMaterial(Spectrum kr, Spectrum kt, Spectrum sa, Spectrum ss, Float scale, Float eta, Float g = 0,
                                          Float uRough = 0, Float vRough = 0);
MaterialDescriptor MakeSubsurfaceMaterial(const char *name, Spectrum kr, Spectrum kt, Float scale, Float eta, Float g = 0,
                                          Float uRough = 0, Float vRough = 0);
MaterialDescriptor MakeSubsurfaceMaterial(const char *name, Float scale, Float eta, Float g = 0, Float uRough = 0,
                                          Float vRough = 0);
MaterialDescriptor MakeKdSubsurfaceMaterial(Spectrum kd, Spectrum kt, Spectrum kr, Spectrum mfp, Float urough, Float vrough,
                                            Float eta, Float scale);
MaterialDescriptor MakePlasticMaterial(Spectrum kd, Spectrum ks, Float rough);
/* The reference's Disney lobes (reflection.h:53-57) behind a material: diffuse (+ fake subsurface when flatness > 0), retro
 * -reflection, sheen and clearcoat, assembled as PBRT-v3's DisneyMaterial does.  The reference's scene API has no such call:
 * its lobes are never pushed. */
MaterialDescriptor MakeDisneyMaterial(Spectrum color, Float roughness, Float sheen = 0, Float sheenTint = .5, Float clearcoat = 0,
                                      Float clearcoatGloss = 1, Float flatness = 0);
MaterialDescriptor MakeUberMaterial(Spectrum kd, Spectrum ks, Spectrum kr, Spectrum kt, Float uRough, Float vRough, Spectrum op,
                                    Float eta);
MaterialDescriptor MakeTranslucentMaterial(Spectrum kd, Spectrum ks, Spectrum reflect, Spectrum transmit, Float rough);
MaterialDescriptor MakePlayGroundMaterial(Spectrum L, int id = -1);
MaterialDescriptor MakePlayGroundMaterial(std::vector<Spectrum> Ls, int id = -1);
MaterialDescriptor MakeEmissive(Spectrum L);

MediumDescriptor MakeMedium(Spectrum sigma_a, Spectrum sigma_s, Float g);
void InsertCameraMedium(MediumDescriptor medium);
/* path-based form kept for API compatibility; needs the EXR reader (row "next") */
void InsertEXRLightMap(const char *path, const Transform &toWorld, const Spectrum &scale);
/* same pipeline (BuildSpectrumMipMap + CreateDistribution2D) from pixels already in memory */
void InsertLightMap(const Spectrum *pixels, int width, int height, const Transform &toWorld, const Spectrum &scale);
/* where that pipeline runs: 0 = on the host (default), 1 = on the current HIP device (ptr_envmap_build: the same tables,
 * bit for bit).  Environment PTR_ENVMAP_BUILD=device selects 1 at start-up.  No counterpart in the reference. */
void SetLightMapBuilder(int on_device);
/* Aggregator::InsertDistantLight(L, w) (primitive.cpp:180-189) as a scene function: the reference defines the light
 * (lights/distant.cpp) and the aggregator method, but nothing in its scene API reaches them.  w: towards the light. */
void InsertDistantLight(const Spectrum &L, const vec3f &w);

void InsertPrimitive(SphereDescriptor shape, MaterialDescriptor mat);
void InsertPrimitive(RectDescriptor shape, MaterialDescriptor mat);
void InsertPrimitive(MeshDescriptor shape, MaterialDescriptor mat);
void InsertPrimitive(BoxDescriptor shape, MaterialDescriptor mat);
void InsertPrimitive(DiskDescriptor shape, MaterialDescriptor mat);
void InsertPrimitive(SphereDescriptor shape, MediumDescriptor medium);
void InsertPrimitive(MeshDescriptor shape, MediumDescriptor medium);
void InsertPrimitive(ParticleCloudDescripor shape, MaterialDescriptor mat);
void InsertPrimitive(SphereDescriptor shape, MaterialDescriptor mat, MediumDescriptor medium);
void InsertPrimitive(MeshDescriptor shape, MaterialDescriptor mat, MediumDescriptor medium);
void InsertPrimitive(ProcToyDescriptor shape, MaterialDescriptor mat);

void PrepareSceneForRendering(Aggregator *scene);
Spectrum SpectrumFromURGB(Float r, Float g, Float b);

/* mesh.cpp:3-26 */
ParsedMesh *ParsedMeshFromData(int nTris, Point3i *indices, int nVerts, Point3f *P);
ParsedMesh *ParsedMeshFromData(const Transform &toWorld, int nTris, Point3i *indices, int nVerts, Point3f *P);

/* ---- render entry points: main.cpp:38-53, 421-457, camera.h:38-42 ----------------------------
 * In the reference SetupPixels and Render are __global__ kernels launched with <<<blocks,threads>>>;
 * here they are host functions with the kernels' argument lists that forward to the C-ABI. */
Image *CreateImage(int res_x, int res_y);
void ImageFree(Image *image);
void SetupPixels(Image *image, unsigned long long seed);
void Render(Image *image, Aggregator *scene, Camera *camera, int ns);
/* integrator selection: the reference edits main.cpp:444-447; default = Li_VolPath as shipped */
void SetIntegrator(int ptr_li_id);
/* copy the device accumulators into image->pixels (the reference reads managed memory) */
void ImageSync(Image *image);

/* ---- film output: camera.h:38-42, ppm.h:8-13, util.h:41 (host/pt_image.cpp) -----------------------
 * The per-pixel tone map runs on the device (ptr_film_rgb8). */
void ImageWrite(Image *image, const char *path, Float exposure, ToneMapAlgorithm algorithm);
void ImageStats(Image *image);
void ImageRGB8(Image *image, Float exposure, ToneMapAlgorithm algorithm, unsigned char *data);
bool ImageWritePPM(Image *image, const char *path, Float exposure, ToneMapAlgorithm algorithm);
bool PPMRead(const char *path, unsigned char **values, int &width, int &height);
void ConvertPPMtoPNG(const char *path, const char *out);
bool EncodePNG(const unsigned char *rgb, int width, int height, bool flip, std::vector<unsigned char> &png);
/* checkpoint / resume of the float accumulators (replaces the temp.ppm backup, graphy.cpp:11,63) */
bool ImageCheckpoint(Image *image, const char *path);
bool ImageResume(Image *image, Aggregator *scene, const char *path);

/* ---- flattening (what PrepareSceneForRendering hands to ptr_scene_upload) -------------------- */
struct FlatSceneHolder;
/* Build the flat description of everything inserted since BeginScene.  The holder owns all
 * arrays the PtrFlatScene points into. */
FlatSceneHolder *FlattenScene();
const PtrFlatScene *FlatSceneGet(const FlatSceneHolder *h);
void FlatSceneFree(FlatSceneHolder *h);
PtrCamera CameraToFlat(const Camera &c);
PtrTransform TransformToFlat(const Transform &t);
