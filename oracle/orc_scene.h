/*
 * orc_scene.h — ORACLE (test infrastructure only; see orc_math.h header).
 * Scene-side data structures of the CPU restatement.  Object graph of the reference
 * (Aggregator -> Primitive -> Shape/Material/Light, pointer-linked BVH Nodes) restated
 * with indices instead of device pointers.  PARITY UNPINNED.
 */
#pragma once
#include "orc_math.h"
#include <vector>

namespace orc {

/* shape.h:8-11 */
enum ShapeType { SPHERE, RECTANGLE, DISK, BOX, MESH, PARTICLECLOUD, SPHERE_PROCEDURAL, BOX_PROCEDURAL, COMPONENT_PROCEDURAL };
/* material.h:250-254 */
enum MaterialType { MatNone = 0, Matte = 1, Mirror, Glass, Metal, Translucent, Plastic, Uber, Subsurface, KdSubsurface, PlayGround, Disney };
/* light.h:13-24 */
enum { LF_DeltaPosition = 1, LF_DeltaDirection = 2, LF_Area = 4, LF_Infinite = 8 };
enum LightType { DiffuseArea = 1, Distant, Infinite };

/* medium.h:41-53 (indices into Scene::media, -1 = nullptr) */
struct MediumInterface {
    int inside, outside;
    MediumInterface() : inside(-1), outside(-1) {}
    explicit MediumInterface(int m) : inside(m), outside(m) {}
    MediumInterface(int i, int o) : inside(i), outside(o) {}
    bool IsMediumTransition() const { return inside != outside; }
};

/* medium.h:10-21 */
struct PhaseFunction {
    Float g;
    int is_initialized;
    PhaseFunction() : g(-2), is_initialized(0) {}
};

/* interaction.h:9-52 */
struct Interaction {
    V3 p;
    V3 pError;
    V3 wo;
    V3 n;
    MediumInterface mediumInterface;
    int faceIndex;
    Interaction() : faceIndex(0) {}
    bool IsSurfaceInteraction() const { return n.x != 0 || n.y != 0 || n.z != 0; } /* n != Normal3f() */
    int GetMedium(const V3 &w) const { return Dot(w, n) > 0 ? mediumInterface.outside : mediumInterface.inside; }
};
/* interaction.h:54-75 */
struct SurfaceInteraction : Interaction {
    V2 uv;
    V3 dpdu, dpdv, dndu, dndv;
    int shape;     /* index of the owning primitive's shape (== primitive index), -1 = nullptr */
    int primitive; /* index into Scene::prims, -1 = nullptr */
    SurfaceInteraction() : shape(-1), primitive(-1) {}
};
/* interaction.h:77-87 */
struct MediumInteraction : Interaction {
    PhaseFunction phase;
    bool IsValid() const { return phase.is_initialized == 1; }
};

/* shape.h:14-25 */
struct PrimitiveHandle {
    Bounds3 bound;
    int handle;
};
struct Node {
    int left, right; /* indices into the node pool, -1 = nullptr */
    int handles;     /* offset into the packed leaf-handle array */
    int n;
    int is_leaf;
    Bounds3 bound;
};
struct BVH {
    std::vector<Node> nodes; /* root = 0 */
    std::vector<PrimitiveHandle> handles;
    int maxLeaf;
};

struct MeshData { /* shape.h:186-196, vertices already in world space (mesh.cpp:61) */
    int nTriangles, nVertices, nUvs;
    std::vector<V3> p;
    std::vector<int> indices; /* 9 per triangle: (pos,nrm,uv) x 3 */
    std::vector<V2> uv;
    bool has_uv;
    BVH bvh;
};
struct CloudData { /* shape.h:232-238 */
    int nParticles;
    Float radius;
    std::vector<V3> positions;
    BVH bvh;
};

struct Shape {
    int type;
    Xf ObjectToWorld, WorldToObject;
    bool reverseOrientation, transformSwapsHandedness;
    /* Sphere (shape.h:58-78) */
    Float radius, thetaMin, thetaMax, phiMax, zMin, zMax;
    /* Rectangle */
    Float sizex, sizey;
    /* Disk */
    Float height, innerRadius;
    /* ProceduralBox */
    V3 length;
    /* ProceduralComponent */
    int toyId;
    Bounds3 toyBounds;
    /* Mesh / ParticleCloud */
    int geom;
};

struct ImageTex {
    int width, height;
    std::vector<V3> data;
};
/* scene.h:52-57 / texture.h:24-84 */
struct Tex {
    int type; /* 0 const spectrum, 1 const float, 2 image */
    V3 spec;
    Float fval;
    int image;
};
struct Material {
    int type;
    Tex tex[16];
    int ntex;
    Float vals[16];
    int id;
    int bssrdfTable; /* index into Scene::tables, -1 */
};

struct Medium { /* medium.h:23-39 */
    V3 sigma_a, sigma_s, sigma_t;
    Float g;
};

struct Primitive { /* primitive.h:13-75 */
    Shape shape;
    MediumInterface mediumInterface;
    int material;    /* index into Scene::materials, -1 = nullptr */
    bool emissive;   /* GeometricEmitterPrimitive */
    V3 L;
    int light;       /* index into Scene::lights for emitters, -1 */
};

/* sampling.h:19-77 */
struct Distribution1D {
    const Float *func, *cdf;
    Float funcInt;
    int size;
};
struct EnvMap {
    int width, height;
    std::vector<V3> texels; /* pyramid level 0 */
    int nu, nv;
    std::vector<Float> func, cdf, funcInt, margCdf;
    Float margInt;
};

struct Light { /* light.h:55-110 */
    int type, flags;
    Xf LightToWorld, WorldToLight;
    V3 Lemit;
    int prim; /* area light: primitive whose shape emits */
    bool twoSided;
    V3 worldCenter;
    Float worldRadius;
    V3 wLight; /* Distant: Normalize(LightToWorld(w)), light.cpp:58-63 */
};

struct BSSRDFTable { /* bssrdf.h:29-42 */
    int nRhoSamples, nRadiusSamples;
    std::vector<Float> rhoSamples, radiusSamples, profile, rhoEff, profileCDF;
    Float EvalProfile(int rhoIndex, int radiusIndex) const { return profile[rhoIndex * nRadiusSamples + radiusIndex]; }
};

struct Counters {
    uint64_t total_paths, misses, surface_hits, medium_hits, light_hits, zero_radiance;
    uint64_t shadow_rays, mis_rays, probe_rays, camera_rays, nodes_visited, prims_tested;
};

extern thread_local Counters g_ctr; /* per-thread; summed by the API layer */

struct Scene { /* primitive.h:77-130 Aggregator */
    std::vector<Primitive> prims;
    std::vector<Material> materials;
    std::vector<Medium> media;
    std::vector<MeshData> meshes;
    std::vector<CloudData> clouds;
    std::vector<ImageTex> images;
    std::vector<Light> lights;
    std::vector<BSSRDFTable> tables;
    BVH top;
    int viewMedium;
    bool has_env;
    EnvMap env;
    int meshfix; /* 1: drop the `*isect = tmp` clobber of mesh.cpp:504-506 (default); 0: as-is */
};

/* glibc rand() restated (TYPE_3 additive feedback generator, r[i] = r[i-3] + r[i-31]) so the
 * BVH axis sequence of primitive.cpp:541 (host rand(), never seeded) is reproducible. */
struct GlibcRand {
    uint32_t buf[31];
    int f, b; /* front / rear ring positions */
    explicit GlibcRand(uint32_t seed = 1);
    int next();
    Float rand_float() { return next() / (2147483647 + 1.f); } /* util.h:64-66 */
};

/* ---- orc_shapes.cpp ---- */
void BuildBVH(BVH &out, std::vector<PrimitiveHandle> &handles, int max_depth, GlibcRand &rng);
Bounds3 ShapeGetBounds(const Scene &sc, const Shape &sh);
bool ShapeIntersect(const Scene &sc, int primIndex, const Ray &ray, Float *tHit, SurfaceInteraction *isect, Float *bary);
Float ShapeArea(const Scene &sc, const Shape &sh);
Interaction ShapeSampleRef(const Scene &sc, const Shape &sh, const Interaction &ref, const V2 &u, Float *pdf);
Float ShapePdfRef(const Scene &sc, int primIndex, const Interaction &ref, const V3 &wi);
bool SceneIntersect(const Scene &sc, const Ray &r, SurfaceInteraction *isect, Float *bary = nullptr);
bool SceneIntersectTr(const Scene &sc, Ray ray, SurfaceInteraction *isect, V3 *Tr);
Ray SpawnRay(const Interaction &it, const V3 &d);
Ray SpawnRayTo(const Interaction &it, const Interaction &to);
Float SDFMapToy(int id, const Bounds3 &bounds, const V3 &p, int *shapeId);

/* ---- orc_bsdf.cpp ---- */
enum BxDFType { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };
enum BxDFImpl { LambertianReflection = 0, LambertianTransmission, SpecularReflection, SpecularTransmission, FresnelSpecular,
                OrenNayar, MicrofacetTransmission, MicrofacetReflection, FresnelBlend, BSSRDFAdapter,
                DisneyDiffuse, DisneyFakeSS, DisneyRetro, DisneySheen, DisneyClearcoat }; /* reflection.h:53-57 */
enum FresnelType { FrNoOp, FrDielectric, FrConductorT };
struct Fresnel { int type; Float etaI, etaT; V3 setaI, setaT, sk; Fresnel() : type(FrNoOp) {} };
struct MicrofacetDistribution { Float alphax, alphay; bool sampleVisibleArea; };
struct SeparableBSSRDF;
struct BxDF {
    int type, impl, is_valid;
    V3 S, T;
    Fresnel fresnel;
    Float A, B;
    V3 sheen;        /* reflection.h:76-80: Disney sheen colour and tint factor */
    Float sheenTint;
    MicrofacetDistribution mDist;
    const SeparableBSSRDF *bssrdf;
    BxDF() : type(0), impl(0), is_valid(1), A(0), B(0), sheenTint(0), bssrdf(nullptr) {}
    bool MatchesFlags(int t) const { return (type & t) == type; }
};
struct SeparableBSSRDF { /* bssrdf.h:84-160 (tabulated only: nothing instantiates Disney) */
    SurfaceInteraction po;
    Float eta;
    V3 ns, ss, ts;
    int material;
    const BSSRDFTable *table;
    V3 sigma_t, rho;
};
struct BSDF { /* reflection.h:358-406 */
    BxDF bxdfs[8];
    SeparableBSSRDF bssrdf;
    int has_bssrdf, nBxDFs;
    V3 ns, ng, ss, ts;
    explicit BSDF(const SurfaceInteraction &si);
    void Push(const BxDF &b) { if (nBxDFs < 8) bxdfs[nBxDFs++] = b; }
    V3 LocalToWorld(const V3 &v) const {
        return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z,
                  ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    V3 WorldToLocal(const V3 &v) const { return V3(Dot(v, ss), Dot(v, ts), Dot(v, ns)); }
    int NumComponents(int flags) const;
    V3 Sample_f(const V3 &woW, V3 *wiW, const V2 &u, Float *pdf, int type, int *sampledType = nullptr) const;
    V3 f(const V3 &woW, const V3 &wiW, int flags = BSDF_ALL) const;
    Float Pdf(const V3 &wo, const V3 &wi, int flags = BSDF_ALL) const;
};
Float FrDieletric(Float cosThetaI, Float etaI, Float etaT);
V3 FrConductor(Float cosThetaI, const V3 &etai, const V3 &etat, const V3 &k);
V3 BxDF_f(const BxDF &b, const V3 &wo, const V3 &wi);
Float BxDF_Pdf(const BxDF &b, const V3 &wo, const V3 &wi);
V3 BxDF_Sample_f(const BxDF &b, const V3 &wo, V3 *wi, const V2 &u, Float *pdf, int *sampledType);
void ComputeScatteringFunctions(const Scene &sc, SurfaceInteraction *si, BSDF *bsdf);
Float RoughnessToAlpha(Float roughness);

/* ---- orc_bssrdf.cpp ---- */
void ComputeBeamDiffusionBSSRDF(Float g, Float eta, BSSRDFTable *t);
void SubsurfaceFromDiffuse(const BSSRDFTable &table, const V3 &rhoEff, const V3 &mfp, V3 *sigma_a, V3 *sigma_s);
void InitTabulatedBSSRDF(SeparableBSSRDF *b, const SurfaceInteraction &po, Float eta, int material, const V3 &sigma_a,
                         const V3 &sigma_s, const BSSRDFTable *table);
V3 BSSRDF_Sw(const SeparableBSSRDF &b, const V3 &w);
V3 BSSRDF_Sample_S(const Scene &sc, const SeparableBSSRDF &b, BSDF *bsdf, Float u1, const V2 &u2, SurfaceInteraction *si,
                   Float *pdf);

/* ---- orc_light.cpp ---- */
V3 LightLe(const Scene &sc, const Light &l, const Ray &r);
V3 LightL(const Light &l, const Interaction &intr, const V3 &w);
V3 SurfaceLe(const Scene &sc, const SurfaceInteraction &si, const V3 &w);
V3 EnvLookup(const EnvMap &env, const V2 &st);
V3 UniformSampleOneLight(const Scene &sc, const Interaction &it, const BSDF *bsdf, const PhaseFunction *phase, V2 u2,
                         V3 u3, bool handleMedium);
V3 MediumTr(const Medium &m, const Ray &ray);
V3 MediumSample(const Scene &sc, int medium, const Ray &ray, const V2 &u, MediumInteraction *mi);
Float PhaseSample_p(const PhaseFunction &ph, const V3 &wo, V3 *wi, const V2 &u);
Float PhaseP(const PhaseFunction &ph, const V3 &wo, const V3 &wi);

/* ---- unit hooks (tests only): the building blocks the known-answer tests integrate, histogram and compare with
 * closed forms (tests/test_cpu_pins.py); thin non-static doors onto the file-local functions ---- */
Float Unit_FrDielectric(Float cosI, Float etaI, Float etaT);
V3 Unit_FrConductor(Float cosI, const V3 &etaI, const V3 &etaT, const V3 &k);
Float Unit_GGX_D(Float ax, Float ay, const V3 &wh);
Float Unit_GGX_G1(Float ax, Float ay, const V3 &w);
Float Unit_GGX_Pdf(Float ax, Float ay, const V3 &wo, const V3 &wh);
V3 Unit_GGX_Sample_wh(Float ax, Float ay, const V3 &wo, const V2 &u);
Float Unit_HG(Float cosTheta, Float g);
V2 Unit_D2_Sample(const EnvMap &e, const V2 &u, Float *pdf);
Float Unit_D2_Pdf(const EnvMap &e, const V2 &p);
V3 Unit_Sr(const SeparableBSSRDF &b, Float r);
Float Unit_Pdf_Sr(const SeparableBSSRDF &b, int ch, Float r);
Float Unit_Sample_Sr(const SeparableBSSRDF &b, int ch, Float u);

/* ---- orc_render.cpp ---- */
struct Camera { V3 position, lower_left, horizontal, vertical, u, v, w; Float lensRadius; };
struct RNG { /* counter-based stand-in for cuRAND (see DESIGN.md "RNG") */
    uint64_t key;
    uint32_t dim;
    Float next();
};
RNG MakeRNG(uint64_t seed, uint32_t pixel, uint32_t sample);
Ray CameraSpawnRay(const Camera &c, Float s, Float t, const V2 &disk, int medium);
V3 Li(const Scene &sc, int integrator, const Ray &ray, RNG &rng);
Float GetFilterWeight(Float u0, Float u1, Float u, Float v);

} // namespace orc
