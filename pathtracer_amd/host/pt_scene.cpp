/*
 * pt_scene.cpp — host side of the kept scene-description / render API (see pt_scene.h).
 * State is file-scope and single-scene, exactly like the reference (hPrimitives, hostScene,
 * scene.cpp:70-95): BeginScene resets it, Make and Insert calls fill it, PrepareSceneForRendering
 * flattens it and uploads through the C-ABI.
 */
#include "pt_scene.h"
#include <cstdlib>
#include <map>

/* ---- transform.cpp ------------------------------------------------------------------------- */
Matrix4x4 Transpose(const Matrix4x4 &m) {
    Matrix4x4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = m.m[j][i];
    return r;
}
/* Gauss-Jordan with full pivoting in Float, the algorithm of transform.cpp:36-90 */
Matrix4x4 Inverse(const Matrix4x4 &m) {
    int indxc[4], indxr[4], ipiv[4] = {0, 0, 0, 0};
    Float a[4][4];
    memcpy(a, m.m, sizeof(a));
    for (int i = 0; i < 4; i++) {
        int irow = 0, icol = 0;
        Float big = 0.f;
        for (int j = 0; j < 4; j++) {
            if (ipiv[j] == 1) continue;
            for (int k = 0; k < 4; k++) {
                if (ipiv[k] == 0) {
                    if (std::abs(a[j][k]) >= big) { big = Float(Absf(a[j][k])); irow = j; icol = k; }
                } else if (ipiv[k] > 1) fprintf(stderr, "Singular matrix in MatrixInvert\n");
            }
        }
        ++ipiv[icol];
        if (irow != icol)
            for (int k = 0; k < 4; ++k) { Float t = a[irow][k]; a[irow][k] = a[icol][k]; a[icol][k] = t; }
        indxr[i] = irow;
        indxc[i] = icol;
        if (a[icol][icol] == 0.f) fprintf(stderr, "Singular matrix in MatrixInvert\n");
        Float pivinv = 1. / a[icol][icol];
        a[icol][icol] = 1.;
        for (int j = 0; j < 4; j++) a[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++) {
            if (j == icol) continue;
            Float save = a[j][icol];
            a[j][icol] = 0;
            for (int k = 0; k < 4; k++) a[j][k] -= a[icol][k] * save;
        }
    }
    for (int j = 3; j >= 0; j--) {
        if (indxr[j] == indxc[j]) continue;
        for (int k = 0; k < 4; k++) { Float t = a[k][indxr[j]]; a[k][indxr[j]] = a[k][indxc[j]]; a[k][indxc[j]] = t; }
    }
    Matrix4x4 r;
    memcpy(r.m, a, sizeof(a));
    return r;
}
Point3f Transform::operator()(const Point3f &p) const {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z + m.m[0][3];
    Float yp = m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z + m.m[1][3];
    Float zp = m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z + m.m[2][3];
    Float wp = m.m[3][0] * x + m.m[3][1] * y + m.m[3][2] * z + m.m[3][3];
    if (wp == 1) return Point3f(xp, yp, zp);
    return Point3f(xp, yp, zp) / wp;
}
vec3f Transform::operator()(const vec3f &v) const {
    Float x = v.x, y = v.y, z = v.z;
    return vec3f(m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z, m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z,
                 m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z);
}
Transform Translate(const vec3f &d) {
    Matrix4x4 m(1, 0, 0, d.x, 0, 1, 0, d.y, 0, 0, 1, d.z, 0, 0, 0, 1);
    Matrix4x4 minv(1, 0, 0, -d.x, 0, 1, 0, -d.y, 0, 0, 1, -d.z, 0, 0, 0, 1);
    return Transform(m, minv);
}
Transform Translate(Float x, Float y, Float z) { return Translate(vec3f(x, y, z)); }
Transform Translate(Float u) { return Translate(vec3f(u, u, u)); }
Transform Scale(Float x, Float y, Float z) {
    Matrix4x4 m(x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1);
    Matrix4x4 minv(1 / x, 0, 0, 0, 0, 1 / y, 0, 0, 0, 0, 1 / z, 0, 0, 0, 0, 1);
    return Transform(m, minv);
}
Transform Scale(Float u) { return Scale(u, u, u); }
Transform RotateX(Float theta) {
    Float s = std::sin(Radians(theta)), c = std::cos(Radians(theta));
    Matrix4x4 m(1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1);
    return Transform(m, Transpose(m));
}
Transform RotateY(Float theta) {
    Float s = std::sin(Radians(theta)), c = std::cos(Radians(theta));
    Matrix4x4 m(c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1);
    return Transform(m, Transpose(m));
}
Transform RotateZ(Float theta) {
    Float s = std::sin(Radians(theta)), c = std::cos(Radians(theta));
    Matrix4x4 m(c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1);
    return Transform(m, Transpose(m));
}
Transform Rotate(Float theta, const vec3f &axis) {
    vec3f a = Normalize(axis);
    Float s = std::sin(Radians(theta)), c = std::cos(Radians(theta));
    Matrix4x4 m;
    m.m[0][0] = a.x * a.x + (1 - a.x * a.x) * c;
    m.m[0][1] = a.x * a.y * (1 - c) - a.z * s;
    m.m[0][2] = a.x * a.z * (1 - c) + a.y * s;
    m.m[1][0] = a.x * a.y * (1 - c) + a.z * s;
    m.m[1][1] = a.y * a.y + (1 - a.y * a.y) * c;
    m.m[1][2] = a.y * a.z * (1 - c) - a.x * s;
    m.m[2][0] = a.x * a.z * (1 - c) - a.y * s;
    m.m[2][1] = a.y * a.z * (1 - c) + a.x * s;
    m.m[2][2] = a.z * a.z + (1 - a.z * a.z) * c;
    return Transform(m, Transpose(m));
}

/* ---- camera.cpp:145-197 ---------------------------------------------------------------------- */
void Camera::Config(Point3f eye, Point3f at, vec3f up, Float fov, Float aspect, Float aperture, Float focus_dist) {
    position = eye;
    lensRadius = aperture / 2;
    Float theta = Radians(fov);
    Float half_height = std::tan(theta / 2.f);
    Float half_width = aspect * half_height;
    w = Normalize(eye - at);
    u = Normalize(Cross(up, w));
    v = Normalize(Cross(w, u));
    lower_left = position - half_width * focus_dist * u - half_height * focus_dist * v - focus_dist * w;
    horizontal = 2.f * half_width * focus_dist * u;
    vertical = 2.f * half_height * focus_dist * v;
}
void Camera::Config(Point3f eye, Point3f at, vec3f up, Float fov, Float aspect, Float aperture) {
    Float focus_dist = (eye - at).Length();
    Config(eye, at, up, fov, aspect, aperture, focus_dist);
}
void Camera::Config(Point3f eye, Point3f at, vec3f up, Float fov, Float aspect) {
    position = eye;
    Float theta = Radians(fov);
    Float half_height = std::tan(theta / 2.f);
    Float half_width = aspect * half_height;
    lensRadius = 0;
    w = Normalize(eye - at);
    u = Normalize(Cross(up, w));
    v = Normalize(Cross(w, u));
    lower_left = position - half_width * u - half_height * v - w;
    horizontal = 2.f * half_width * u;
    vertical = 2.f * half_height * v;
}

/* ---- scene state (scene.cpp:70-95) ------------------------------------------------------------- */
struct LightMapState {
    bool valid;
    Transform toWorld;
    int width, height; /* level 0, powers of two */
    std::vector<Spectrum> texels;
    int nu, nv;
    std::vector<Float> func, cdf, funcInt, margCdf;
    Float margInt;
};
static std::vector<PrimitiveDescriptor> hPrimitives;
static MediumDescriptor g_cameraMedium = {Spectrum(0), Spectrum(0), 0, 0};
static LightMapState g_lightMap;
static int mplayground_ids = 0;
static int g_integrator = PTR_LI_VOLPATH; /* main.cpp:447 as shipped */
static PtrContext *g_default_ctx = nullptr;

static void die_on(int rc, const char *what) { /* cudaDeviceAssert semantics: print and exit */
    if (rc != PTR_OK) {
        fprintf(stderr, "[pathtracer_amd] %s failed: %s\n", what, ptr_last_error());
        exit(1);
    }
}
static PtrContext *default_ctx() {
    if (!g_default_ctx) die_on(ptr_create(-1, &g_default_ctx), "ptr_create");
    return g_default_ctx;
}

static std::vector<PtrDistantLight> g_distantLights;
void InsertDistantLight(const Spectrum &L, const vec3f &w) {
    PtrDistantLight d;
    d.L[0] = L.x; d.L[1] = L.y; d.L[2] = L.z;
    d.w[0] = w.x; d.w[1] = w.y; d.w[2] = w.z;
    g_distantLights.push_back(d);
}
void BeginScene(Aggregator *scene) {
    hPrimitives.clear();
    g_cameraMedium.is_valid = 0;
    g_lightMap.valid = false;
    g_distantLights.clear();
    if (scene && !scene->ctx) scene->ctx = default_ctx();
}

/* ---- descriptors, scene.cpp:130-229 ----------------------------------------------------------------- */
TextureDescriptor MakeTexture(Spectrum value) { TextureDescriptor d = {}; d.type = 0; d.spec_val = value; return d; }
TextureDescriptor MakeTexture(Float value) { TextureDescriptor d = {}; d.type = 1; d.fval = value; return d; }
TextureDescriptor MakeTexture(ImageData *data) { TextureDescriptor d = {}; d.type = 2; d.image = data; return d; }

SphereDescriptor MakeSphereProcedural(Transform toWorld, Float radius) {
    SphereDescriptor d; d.toWorld = toWorld; d.radius = radius; d.procedural = true; d.reverseOrientation = false; return d;
}
SphereDescriptor MakeSphere(Transform toWorld, Float radius, bool rev) {
    SphereDescriptor d; d.toWorld = toWorld; d.radius = radius; d.reverseOrientation = rev; d.procedural = false; return d;
}
BoxDescriptor MakeBox(Transform toWorld, Float sx, Float sy, Float sz, bool rev) {
    BoxDescriptor d; d.toWorld = toWorld; d.sizex = sx; d.sizey = sy; d.sizez = sz; d.reverseOrientation = rev; return d;
}
RectDescriptor MakeRectangle(Transform toWorld, Float sx, Float sy, bool rev) {
    RectDescriptor d; d.toWorld = toWorld; d.sizex = sx; d.sizey = sy; d.reverseOrientation = rev; return d;
}
DiskDescriptor MakeDisk(Transform toWorld, Float height, Float radius, Float innerRadius, Float phiMax, bool rev) {
    DiskDescriptor d; d.toWorld = toWorld; d.height = height; d.radius = radius; d.innerRadius = innerRadius; d.phiMax = phiMax;
    d.reverseOrientation = rev; return d;
}
MeshDescriptor MakeMesh(ParsedMesh *mesh) { MeshDescriptor d; d.mesh = mesh; return d; }
ParticleCloudDescripor MakeParticleCloud(vec3f *positions, int n, Float size) {
    ParticleCloudDescripor d; d.pos = positions; d.npos = n; d.scale = size; return d;
}
ProcToyDescriptor MakeProceduralToy(Transform toWorld, Bounds3f bound, int id) {
    ProcToyDescriptor d; d.toWorld = toWorld; d.bound = bound; d.id = id; return d;
}

/* ---- materials, scene.cpp:231-482.  Unlike the reference every descriptor starts zeroed
 * (specs_taken/id/vals and unused texture slots are read uninitialised there). ---- */
static MaterialDescriptor blank(MaterialType t) {
    MaterialDescriptor d;
    memset((void *)&d, 0, sizeof(d));
    d.type = t;
    return d;
}
MaterialDescriptor MakeMatteMaterial(Spectrum kd, Float sigma) {
    MaterialDescriptor d = blank(Matte); d.textures[0] = MakeTexture(kd); d.textures[1] = MakeTexture(sigma); d.specs_taken = 2; return d;
}
MaterialDescriptor MakeMatteMaterial(TextureDescriptor kd, Float sigma) {
    MaterialDescriptor d = blank(Matte); d.textures[0] = kd; d.textures[1] = MakeTexture(sigma); d.specs_taken = 2; return d;
}
MaterialDescriptor MakePlayGroundMaterial(Spectrum L, int id) {
    MaterialDescriptor d = blank(PlayGround); d.textures[0] = MakeTexture(L); d.specs_taken = 1;
    d.id = id > -1 ? id : mplayground_ids; mplayground_ids++; return d;
}
MaterialDescriptor MakePlayGroundMaterial(std::vector<Spectrum> Ls, int id) {
    MaterialDescriptor d = blank(PlayGround);
    size_t n = Ls.size() < 16 ? Ls.size() : 16;
    for (size_t i = 0; i < n; i++) d.textures[i] = MakeTexture(Ls[i]);
    d.id = id > -1 ? id : mplayground_ids; d.specs_taken = (int)n; mplayground_ids++; return d;
}
static MaterialDescriptor glass5(TextureDescriptor kr, TextureDescriptor kt, TextureDescriptor ur, TextureDescriptor vr, Float index) {
    MaterialDescriptor d = blank(Glass);
    d.textures[0] = kr; d.textures[1] = kt; d.textures[2] = ur; d.textures[3] = vr; d.textures[4] = MakeTexture(index);
    d.specs_taken = 5; return d;
}
MaterialDescriptor MakeGlassMaterial(TextureDescriptor kr, TextureDescriptor kt, Float index, TextureDescriptor rough) {
    return glass5(kr, kt, rough, rough, index);
}
MaterialDescriptor MakeGlassMaterial(Spectrum kr, Spectrum kt, Float index, TextureDescriptor rough) {
    return glass5(MakeTexture(kr), MakeTexture(kt), rough, rough, index);
}
MaterialDescriptor MakeGlassMaterial(Spectrum kr, Spectrum kt, Float index, Float uRough, Float vRough) {
    return glass5(MakeTexture(kr), MakeTexture(kt), MakeTexture(uRough), MakeTexture(vRough), index);
}
MaterialDescriptor MakeTranslucentMaterial(Spectrum kd, Spectrum ks, Spectrum reflect, Spectrum transmit, Float rough) {
    MaterialDescriptor d = blank(Translucent);
    d.textures[0] = MakeTexture(kd); d.textures[1] = MakeTexture(ks); d.textures[2] = MakeTexture(reflect);
    d.textures[3] = MakeTexture(transmit); d.textures[4] = MakeTexture(rough); d.specs_taken = 5; return d;
}
MaterialDescriptor MakeKdSubsurfaceMaterial(Spectrum kd, Spectrum kt, Spectrum kr, Spectrum mfp, Float urough, Float vrough, Float eta,
                                            Float scale) {
    MaterialDescriptor d = blank(KdSubsurface);
    d.textures[0] = MakeTexture(kd); d.textures[1] = MakeTexture(kt); d.textures[2] = MakeTexture(kr); d.textures[3] = MakeTexture(mfp);
    d.textures[4] = MakeTexture(urough); d.textures[5] = MakeTexture(vrough); d.vals[0] = eta; d.vals[1] = scale; d.specs_taken = 6;
    return d;
}
struct MeasuredSS { const char *name; Float sigma_prime_s[3], sigma_a[3]; };
/* Published measurements (Jensen et al. 2001; Narasimhan et al. 2006) as tabulated in
 * scene.cpp:12-68; the entries the shipped scenes name plus the milk/skin/marble set. */
static const MeasuredSS kSubsurfaceTable[] = {
    {"Apple", {2.29, 2.39, 1.97}, {0.0030, 0.0034, 0.046}},
    {"Chicken1", {0.15, 0.21, 0.38}, {0.015, 0.077, 0.19}},
    {"Chicken2", {0.19, 0.25, 0.32}, {0.018, 0.088, 0.20}},
    {"Cream", {7.38, 5.47, 3.15}, {0.0002, 0.0028, 0.0163}},
    {"Ketchup", {0.18, 0.07, 0.03}, {0.061, 0.97, 1.45}},
    {"Marble", {2.19, 2.62, 3.00}, {0.0021, 0.0041, 0.0071}},
    {"Potato", {0.68, 0.70, 0.55}, {0.0024, 0.0090, 0.12}},
    {"Skimmilk", {0.70, 1.22, 1.90}, {0.0014, 0.0025, 0.0142}},
    {"Skin1", {0.74, 0.88, 1.01}, {0.032, 0.17, 0.48}},
    {"Skin2", {1.09, 1.59, 1.79}, {0.013, 0.070, 0.145}},
    {"Spectralon", {11.6, 20.4, 14.9}, {0.00, 0.00, 0.00}},
    {"Wholemilk", {2.55, 3.21, 3.77}, {0.0011, 0.0024, 0.014}},
    {"Coke", {8.9053e-05, 8.372e-05, 0}, {0.10014, 0.16503, 0.2468}},
    {"Pepsi", {6.1697e-05, 4.2564e-05, 0}, {0.091641, 0.14158, 0.20729}},
    {"Espresso", {0.72378, 0.84557, 1.0247}, {4.7984, 6.5751, 8.8493}},
    {"Regular Milk", {4.5513, 5.8294, 7.136}, {0.0015333, 0.0046, 0.019933}},
};
static bool GetMediumScatteringProperties(const std::string &name, Spectrum *sigma_a, Spectrum *sigma_prime_s) {
    for (const MeasuredSS &m : kSubsurfaceTable)
        if (name == m.name) {
            *sigma_a = Spectrum(m.sigma_a[0], m.sigma_a[1], m.sigma_a[2]);
            *sigma_prime_s = Spectrum(m.sigma_prime_s[0], m.sigma_prime_s[1], m.sigma_prime_s[2]);
            return true;
        }
    return false;
}
MaterialDescriptor MakeSubsurfaceMaterial(Spectrum kr, Spectrum kt, Spectrum sa, Spectrum ss, Float scale, Float eta, Float g,
                                          Float uRough, Float vRough) {
    MaterialDescriptor d = blank(Subsurface);
    d.textures[0] = MakeTexture(kr); d.textures[1] = MakeTexture(kt); d.textures[2] = MakeTexture(sa); d.textures[3] = MakeTexture(ss);
    d.textures[4] = MakeTexture(uRough); d.textures[5] = MakeTexture(vRough);
    d.vals[0] = scale; d.vals[1] = eta; d.vals[2] = g; d.specs_taken = 6; return d;
}
MaterialDescriptor MakeSubsurfaceMaterial(const char *name, Spectrum kr, Spectrum kt, Float scale, Float eta, Float g, Float uRough,
                                          Float vRough) {
    Spectrum sig_a(.0011f, .0024f, .014f), sig_s(2.55f, 3.21f, 3.77f);
    if (!GetMediumScatteringProperties(name, &sig_a, &sig_s))
        printf("Warning: Could not find scattering properties for %s, using default\n", name);
    return MakeSubsurfaceMaterial(kr, kt, sig_a, sig_s, scale, eta, g, uRough, vRough);
}
MaterialDescriptor MakeSubsurfaceMaterial(const char *name, Float scale, Float eta, Float g, Float uRough, Float vRough) {
    return MakeSubsurfaceMaterial(name, Spectrum(1), Spectrum(1), scale, eta, g, uRough, vRough);
}
MaterialDescriptor MakeMirrorMaterial(Spectrum kr) { MaterialDescriptor d = blank(Mirror); d.textures[0] = MakeTexture(kr); d.specs_taken = 1; return d; }
MaterialDescriptor MakeMetalMaterial(Spectrum kr, Spectrum k, Float etaI, Float etaT) {
    MaterialDescriptor d = blank(Metal);
    d.textures[0] = MakeTexture(kr); d.textures[1] = MakeTexture(k); d.textures[2] = MakeTexture(etaI); d.textures[3] = MakeTexture(etaT);
    d.specs_taken = 4; return d;
}
MaterialDescriptor MakePlasticMaterial(Spectrum kd, Spectrum ks, Float rough) {
    MaterialDescriptor d = blank(Plastic);
    d.textures[0] = MakeTexture(kd); d.textures[1] = MakeTexture(ks); d.textures[2] = MakeTexture(rough); d.specs_taken = 3; return d;
}
MaterialDescriptor MakeDisneyMaterial(Spectrum color, Float roughness, Float sheen, Float sheenTint, Float clearcoat, Float clearcoatGloss,
                                      Float flatness) {
    MaterialDescriptor d = blank(Disney);
    d.textures[0] = MakeTexture(color); d.textures[1] = MakeTexture(roughness); d.textures[2] = MakeTexture(sheen);
    d.textures[3] = MakeTexture(sheenTint); d.textures[4] = MakeTexture(clearcoat); d.textures[5] = MakeTexture(clearcoatGloss);
    d.textures[6] = MakeTexture(flatness); d.specs_taken = 7; return d;
}
MaterialDescriptor MakeUberMaterial(Spectrum kd, Spectrum ks, Spectrum kr, Spectrum kt, Float uRough, Float vRough, Spectrum op,
                                    Float eta) {
    MaterialDescriptor d = blank(Uber);
    d.textures[0] = MakeTexture(kd); d.textures[1] = MakeTexture(ks); d.textures[2] = MakeTexture(kr); d.textures[3] = MakeTexture(kt);
    d.textures[4] = MakeTexture(uRough); d.textures[5] = MakeTexture(vRough); d.textures[6] = MakeTexture(op);
    d.textures[7] = MakeTexture(eta); d.specs_taken = 8; return d;
}
MaterialDescriptor MakeEmissive(Spectrum L) {
    MaterialDescriptor d = blank((MaterialType)0); d.is_emissive = 1; d.textures[0] = MakeTexture(L); d.specs_taken = 1; return d;
}
MediumDescriptor MakeMedium(Spectrum sigma_a, Spectrum sigma_s, Float g) {
    MediumDescriptor d; d.sa = sigma_a; d.ss = sigma_s; d.g = g; d.is_valid = 1; return d;
}
Spectrum SpectrumFromURGB(Float r, Float g, Float b) { return Spectrum(r / 255.f, g / 255.f, b / 255.f); }

/* ---- InsertPrimitive overloads, scene.cpp:588-701 ------------------------------------------------- */
static PrimitiveDescriptor blank_prim(ShapeType t) {
    PrimitiveDescriptor d;
    memset((void *)&d, 0, sizeof(d));
    d.shapeType = t;
    return d;
}
static MediumDescriptor no_medium() { MediumDescriptor m; m.sa = Spectrum(0); m.ss = Spectrum(0); m.g = 0; m.is_valid = 0; return m; }
void InsertPrimitive(ProcToyDescriptor s, MaterialDescriptor mat) { PrimitiveDescriptor d = blank_prim(COMPONENT_PROCEDURAL); d.mat = mat; d.toyDesc = s; hPrimitives.push_back(d); }
void InsertPrimitive(ParticleCloudDescripor s, MaterialDescriptor mat) { PrimitiveDescriptor d = blank_prim(PARTICLECLOUD); d.partDesc = s; d.mat = mat; hPrimitives.push_back(d); }
void InsertPrimitive(SphereDescriptor s, MaterialDescriptor mat, MediumDescriptor medium) {
    PrimitiveDescriptor d = blank_prim(SPHERE); d.mat = mat; d.mediumDesc = medium; d.sphereDesc = s; hPrimitives.push_back(d);
}
void InsertPrimitive(SphereDescriptor s, MaterialDescriptor mat) { InsertPrimitive(s, mat, no_medium()); }
void InsertPrimitive(BoxDescriptor s, MaterialDescriptor mat) { PrimitiveDescriptor d = blank_prim(BOX_PROCEDURAL); d.mat = mat; d.boxDesc = s; hPrimitives.push_back(d); }
void InsertPrimitive(RectDescriptor s, MaterialDescriptor mat) { PrimitiveDescriptor d = blank_prim(RECTANGLE); d.mat = mat; d.rectDesc = s; hPrimitives.push_back(d); }
void InsertPrimitive(MeshDescriptor s, MaterialDescriptor mat, MediumDescriptor medium) {
    PrimitiveDescriptor d = blank_prim(MESH); d.mat = mat; d.mediumDesc = medium; d.meshDesc = s; hPrimitives.push_back(d);
}
void InsertPrimitive(MeshDescriptor s, MaterialDescriptor mat) { InsertPrimitive(s, mat, no_medium()); }
void InsertPrimitive(DiskDescriptor s, MaterialDescriptor mat) { PrimitiveDescriptor d = blank_prim(DISK); d.mat = mat; d.diskDesc = s; hPrimitives.push_back(d); }
void InsertPrimitive(SphereDescriptor s, MediumDescriptor medium) {
    PrimitiveDescriptor d = blank_prim(SPHERE); d.no_mat = 1; d.sphereDesc = s; d.mediumDesc = medium; hPrimitives.push_back(d);
}
void InsertPrimitive(MeshDescriptor s, MediumDescriptor medium) {
    PrimitiveDescriptor d = blank_prim(MESH); d.no_mat = 1; d.meshDesc = s; d.mediumDesc = medium; hPrimitives.push_back(d);
}
void InsertCameraMedium(MediumDescriptor medium) { g_cameraMedium = medium; }

ParsedMesh *ParsedMeshFromData(const Transform &toWorld, int nTris, Point3i *indices, int nVerts, Point3f *P) {
    ParsedMesh *mesh = new ParsedMesh();
    mesh->nUvs = 0; mesh->nNormals = 0; mesh->s = nullptr; mesh->uv = nullptr; mesh->n = nullptr;
    mesh->toWorld = toWorld;
    mesh->nVertices = nVerts;
    mesh->nTriangles = nTris;
    mesh->p = (Point3f *)malloc(sizeof(Point3f) * nVerts);
    mesh->indices = (Point3i *)malloc(sizeof(Point3i) * 3 * (size_t)nTris);
    memcpy((void *)mesh->p, P, sizeof(Point3f) * nVerts);
    memcpy((void *)mesh->indices, indices, sizeof(Point3i) * 3 * (size_t)nTris);
    return mesh;
}
ParsedMesh *ParsedMeshFromData(int nTris, Point3i *indices, int nVerts, Point3f *P) {
    return ParsedMeshFromData(Translate(0, 0, 0), nTris, indices, nVerts, P);
}

/* ---- environment map: texture.cpp:19-27,91-260, infinite.cpp:7-39 (setup, host) --------------------- */
static Float Lanczos(Float x, Float tau = 2) {
    x = std::abs(x);
    if (x < 1e-5f) return 1;
    if (x > 1.f) return 0;
    x *= Pi;
    Float s = std::sin(x * tau) / (x * tau);
    Float lanczos = std::sin(x) / x;
    return s * lanczos;
}
static int ModI(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }
static bool IsPow2(int v) { return v && !(v & (v - 1)); }
static int RoundUpPow2(int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
static int Log2Int(unsigned v) { int i = -1; while (v) { v >>= 1; i++; } return i; }
struct ResampleWeight { int firstTexel; Float weight[4]; };
static std::vector<ResampleWeight> GetResampleWeights(int oldRes, int newRes) {
    std::vector<ResampleWeight> wt(newRes);
    Float filterwidth = 2.f;
    for (int i = 0; i < newRes; i++) {
        Float center = (i + .5f) * oldRes / newRes;
        wt[i].firstTexel = std::floor((center - filterwidth) + 0.5f);
        for (int j = 0; j < 4; j++) {
            Float pos = wt[i].firstTexel + j + .5f;
            wt[i].weight[j] = Lanczos((pos - center) / filterwidth);
        }
        Float invSumWts = 1 / (wt[i].weight[0] + wt[i].weight[1] + wt[i].weight[2] + wt[i].weight[3]);
        for (int j = 0; j < 4; ++j) wt[i].weight[j] *= invSumWts;
    }
    return wt;
}
struct Level { std::vector<Spectrum> t; int rx, ry; };
static const Spectrum &LTexel(const Level &l, int s, int t) { return l.t[ModI(s, l.rx) + ModI(t, l.ry) * l.rx]; }
static Spectrum LTriangle(const std::vector<Level> &pyr, int level, Float st0, Float st1) {
    if (level < 0) level = 0;
    if (level > (int)pyr.size() - 1) level = (int)pyr.size() - 1;
    const Level &l = pyr[level];
    Float s = st0 * l.rx - 0.5f, t = st1 * l.ry - 0.5f;
    int s0 = std::floor(s), t0 = std::floor(t);
    Float ds = s - s0, dt = t - t0;
    return (1 - ds) * (1 - dt) * LTexel(l, s0, t0) + (1 - ds) * dt * LTexel(l, s0, t0 + 1) + ds * (1 - dt) * LTexel(l, s0 + 1, t0) +
           ds * dt * LTexel(l, s0 + 1, t0 + 1);
}
static Spectrum LLookup(const std::vector<Level> &pyr, Float st0, Float st1, Float width) {
    int levels = (int)pyr.size();
    const Float invLog2 = 1.442695040888963387004650940071;
    Float level = levels - 1 + std::log(Max(width, (Float)1e-8)) * invLog2;
    if (level < 0) return LTriangle(pyr, 0, st0, st1);
    if (level >= levels - 1) return LTexel(pyr[levels - 1], 0, 0);
    int iLevel = std::floor(level);
    Float delta = level - iLevel;
    return (1 - delta) * LTriangle(pyr, iLevel, st0, st1) + delta * LTriangle(pyr, iLevel + 1, st0, st1);
}
static void Distribution1D(const Float *f, int n, Float *cdf, Float *funcInt) { /* infinite.cpp:7-22 */
    cdf[0] = 0;
    for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + f[i - 1] / n;
    *funcInt = cdf[n];
    if (*funcInt == 0) for (int i = 1; i < n + 1; ++i) cdf[i] = Float(i) / Float(n);
    else for (int i = 1; i < n + 1; ++i) cdf[i] /= *funcInt;
}
static int g_lightmap_on_device = -1;
void SetLightMapBuilder(int on_device) { g_lightmap_on_device = on_device ? 1 : 0; }
static bool lightmap_on_device() {
    if (g_lightmap_on_device < 0) {
        const char *e = getenv("PTR_ENVMAP_BUILD");
        g_lightmap_on_device = (e && !strcmp(e, "device")) ? 1 : 0;
    }
    return g_lightmap_on_device == 1;
}
void InsertLightMap(const Spectrum *pixels, int width, int height, const Transform &toWorld, const Spectrum &scale) {
    if (g_lightMap.valid) { printf("Warning: Multiple calls to InsertLightMap\n"); return; }
    if (lightmap_on_device()) { /* the device pipeline (csrc/rt_envmap.hip) */
        static_assert(sizeof(Spectrum) == 3 * sizeof(float), "Spectrum is three floats");
        const float sc3[3] = {scale.x, scale.y, scale.z};
        PtrEnvTables *tables = nullptr;
        if (ptr_envmap_build(-1, reinterpret_cast<const float *>(pixels), width, height, sc3, &tables) != PTR_OK) {
            fprintf(stderr, "[pathtracer_amd] InsertLightMap: %s\n", ptr_last_error());
            exit(1);
        }
        PtrEnvMap v;
        memset(&v, 0, sizeof(v));
        ptr_envmap_view(tables, &v, nullptr);
        LightMapState &lm = g_lightMap;
        lm.toWorld = toWorld;
        lm.width = v.width; lm.height = v.height; lm.nu = v.nu; lm.nv = v.nv;
        lm.texels.resize((size_t)v.width * v.height);
        memcpy((void *)lm.texels.data(), v.texels, sizeof(float) * 3 * (size_t)v.width * v.height);
        lm.func.assign(v.func, v.func + (size_t)v.nu * v.nv);
        lm.cdf.assign(v.cdf, v.cdf + (size_t)(v.nu + 1) * v.nv);
        lm.funcInt.assign(v.func_int, v.func_int + v.nv);
        lm.margCdf.assign(v.marg_cdf, v.marg_cdf + v.nv + 1);
        lm.margInt = v.marg_int;
        lm.valid = true;
        ptr_envmap_free(tables);
        return;
    }
    std::vector<Spectrum> L(pixels, pixels + (size_t)width * height);
    for (auto &p : L) p = p * scale;
    int rx = width, ry = height;
    Level l0;
    if (!IsPow2(rx) || !IsPow2(ry)) { /* texture.cpp:117-172, wrap = Repeat */
        int px = RoundUpPow2(rx), py = RoundUpPow2(ry);
        std::vector<Spectrum> res((size_t)px * py);
        std::vector<ResampleWeight> sW = GetResampleWeights(rx, px);
        for (int t = 0; t < ry; t++)
            for (int s = 0; s < px; s++) {
                Spectrum acc(0.f);
                for (int j = 0; j < 4; j++) {
                    int origS = ModI(sW[s].firstTexel + j, rx);
                    acc = acc + sW[s].weight[j] * L[(size_t)t * rx + origS];
                }
                res[(size_t)t * px + s] = acc;
            }
        std::vector<ResampleWeight> tW = GetResampleWeights(ry, py);
        std::vector<Spectrum> work(py);
        for (int s = 0; s < px; s++) {
            for (int t = 0; t < py; t++) {
                Spectrum acc(0.f);
                for (int j = 0; j < 4; j++) {
                    int offset = ModI(tW[t].firstTexel + j, ry);
                    acc = acc + tW[t].weight[j] * res[(size_t)offset * px + s];
                }
                work[t] = acc;
            }
            for (int t = 0; t < py; t++) {
                Spectrum w = work[t]; /* Clamp(workData[t], T(0), T(1)) */
                for (int c = 0; c < 3; c++) w[c] = w[c] < 0 ? 0 : (w[c] > 1 ? 1 : w[c]);
                res[(size_t)t * px + s] = w;
            }
        }
        l0.t.swap(res); l0.rx = px; l0.ry = py;
    } else { l0.t = L; l0.rx = rx; l0.ry = ry; }
    std::vector<Level> pyr;
    pyr.push_back(l0);
    int nLevels = 1 + Log2Int((unsigned)(l0.rx > l0.ry ? l0.rx : l0.ry));
    for (int i = 1; i < nLevels; i++) { /* texture.cpp:189-212 */
        const Level &p = pyr[i - 1];
        Level c;
        c.rx = p.rx / 2 > 1 ? p.rx / 2 : 1;
        c.ry = p.ry / 2 > 1 ? p.ry / 2 : 1;
        c.t.resize((size_t)c.rx * c.ry);
        for (int t = 0; t < c.ry; t++)
            for (int s = 0; s < c.rx; s++)
                c.t[s + (size_t)t * c.rx] =
                    0.25f * (LTexel(p, 2 * s, 2 * t) + LTexel(p, 2 * s + 1, 2 * t) + LTexel(p, 2 * s, 2 * t + 1) + LTexel(p, 2 * s + 1, 2 * t + 1));
        pyr.push_back(c);
    }
    LightMapState &lm = g_lightMap;
    lm.toWorld = toWorld;
    lm.width = l0.rx; lm.height = l0.ry;
    lm.texels = pyr[0].t;
    int W = 2 * lm.width, H = 2 * lm.height; /* texture.cpp:235-252 */
    lm.nu = W; lm.nv = H;
    lm.func.resize((size_t)W * H);
    float fwidth = 0.5f / Min(W, H);
    for (int v = 0; v < H; v++) {
        Float vp = (v + .5f) / (Float)H;
        Float sinTheta = std::sin((Float)(Pi * (v + .5f) / H));
        for (int u = 0; u < W; u++) {
            Float up = (u + .5f) / (Float)W;
            Spectrum s = LLookup(pyr, up, vp, fwidth);
            Float y = 0.212671f * s.x + 0.715160f * s.y + 0.072169f * s.z;
            lm.func[u + (size_t)v * W] = y;
            lm.func[u + (size_t)v * W] *= sinTheta;
        }
    }
    lm.cdf.resize((size_t)(W + 1) * H);
    lm.funcInt.resize(H);
    for (int v = 0; v < H; v++) Distribution1D(&lm.func[(size_t)v * W], W, &lm.cdf[(size_t)v * (W + 1)], &lm.funcInt[v]);
    lm.margCdf.resize(H + 1);
    Distribution1D(lm.funcInt.data(), H, lm.margCdf.data(), &lm.margInt);
    lm.valid = true;
}
/* scene.cpp:703-719 + texture.cpp:217-257: read, scale, MipMap, Distribution2D */
Spectrum *ReadImageEXR(const char *path, int &width, int &height); /* pt_loaders.cpp */
void InsertEXRLightMap(const char *path, const Transform &toWorld, const Spectrum &scale) {
    if (g_lightMap.valid) { printf("Warning: Multiple calls to InsertLightMap\n"); return; }
    printf("Generating environment map\n * Reading EXR Map...");
    int width = 0, height = 0;
    Spectrum *L = ReadImageEXR(path, width, height);
    if (!L) { /* the reference asserts on a null image */
        fprintf(stderr, "[pathtracer_amd] InsertEXRLightMap(%s): could not read the image\n", path);
        exit(1);
    }
    printf("OK\n");
    InsertLightMap(L, width, height, toWorld, scale);
    delete[] L;
}

/* ---- flattening -------------------------------------------------------------------------------------- */
struct FlatSceneHolder {
    PtrFlatScene fs;
    std::vector<PtrPrimitive> prims;
    std::vector<PtrMesh> meshes;
    std::vector<PtrCloud> clouds;
    std::vector<PtrImage> images;
    std::vector<PtrDistantLight> distant;
    LightMapState lm;
};
PtrTransform TransformToFlat(const Transform &t) {
    PtrTransform o;
    memcpy(o.m, t.m.m, sizeof(o.m));
    memcpy(o.mInv, t.mInv.m, sizeof(o.mInv));
    return o;
}
PtrCamera CameraToFlat(const Camera &c) {
    PtrCamera o;
    memcpy(o.position, &c.position, 12); memcpy(o.lower_left, &c.lower_left, 12);
    memcpy(o.horizontal, &c.horizontal, 12); memcpy(o.vertical, &c.vertical, 12);
    o.lens_radius = c.lensRadius;
    memcpy(o.u, &c.u, 12); memcpy(o.v, &c.v, 12); memcpy(o.w, &c.w, 12);
    return o;
}
static PtrMedium flat_medium(const MediumDescriptor &m) {
    PtrMedium o;
    memcpy(o.sigma_a, &m.sa, 12); memcpy(o.sigma_s, &m.ss, 12);
    o.g = m.g; o.is_valid = m.is_valid;
    return o;
}
FlatSceneHolder *FlattenScene() {
    FlatSceneHolder *h = new FlatSceneHolder();
    std::map<const ParsedMesh *, int> meshIdx;
    std::map<const vec3f *, int> cloudIdx;
    std::map<const ImageData *, int> imageIdx;
    for (const PrimitiveDescriptor &pd : hPrimitives) {
        PtrPrimitive p;
        memset(&p, 0, sizeof(p));
        p.shape_type = pd.shapeType;
        p.no_mat = pd.no_mat;
        p.geom = -1;
        Transform tw;
        switch (pd.shapeType) {
        case SPHERE:
            tw = pd.sphereDesc.toWorld; p.params[0] = pd.sphereDesc.radius; p.reverse_orientation = pd.sphereDesc.reverseOrientation;
            if (pd.sphereDesc.procedural) p.shape_type = SPHERE_PROCEDURAL; /* scene.cpp:733-739 */
            break;
        case RECTANGLE:
            tw = pd.rectDesc.toWorld; p.params[0] = pd.rectDesc.sizex; p.params[1] = pd.rectDesc.sizey;
            p.reverse_orientation = pd.rectDesc.reverseOrientation;
            break;
        case DISK:
            tw = pd.diskDesc.toWorld; p.params[0] = pd.diskDesc.height; p.params[1] = pd.diskDesc.radius;
            p.params[2] = pd.diskDesc.innerRadius; p.params[3] = pd.diskDesc.phiMax; p.reverse_orientation = pd.diskDesc.reverseOrientation;
            break;
        case BOX_PROCEDURAL:
            tw = pd.boxDesc.toWorld; p.params[0] = pd.boxDesc.sizex; p.params[1] = pd.boxDesc.sizey; p.params[2] = pd.boxDesc.sizez;
            break;
        case COMPONENT_PROCEDURAL:
            tw = pd.toyDesc.toWorld;
            p.params[0] = pd.toyDesc.bound.pMin.x; p.params[1] = pd.toyDesc.bound.pMin.y; p.params[2] = pd.toyDesc.bound.pMin.z;
            p.params[3] = pd.toyDesc.bound.pMax.x; p.params[4] = pd.toyDesc.bound.pMax.y; p.params[5] = pd.toyDesc.bound.pMax.z;
            p.geom = pd.toyDesc.id;
            break;
        case MESH: {
            const ParsedMesh *pm = pd.meshDesc.mesh;
            auto it = meshIdx.find(pm);
            if (it == meshIdx.end()) {
                PtrMesh m;
                memset(&m, 0, sizeof(m));
                m.n_triangles = pm->nTriangles; m.n_vertices = pm->nVertices; m.n_uvs = pm->uv ? pm->nUvs : 0;
                m.p = (const float *)pm->p; m.indices = (const int32_t *)pm->indices; m.uv = (const float *)pm->uv;
                m.to_world = TransformToFlat(pm->toWorld);
                it = meshIdx.insert({pm, (int)h->meshes.size()}).first;
                h->meshes.push_back(m);
            }
            p.geom = it->second;
            tw = pm->toWorld;
        } break;
        case PARTICLECLOUD: {
            auto it = cloudIdx.find(pd.partDesc.pos);
            if (it == cloudIdx.end()) {
                PtrCloud c;
                c.n_particles = pd.partDesc.npos; c.radius = pd.partDesc.scale; c.positions = (const float *)pd.partDesc.pos;
                it = cloudIdx.insert({pd.partDesc.pos, (int)h->clouds.size()}).first;
                h->clouds.push_back(c);
            }
            p.geom = it->second;
            p.params[0] = pd.partDesc.scale;
        } break;
        default: break;
        }
        p.to_world = TransformToFlat(tw);
        PtrMaterial &m = p.mat;
        m.type = pd.mat.is_emissive ? PTR_MAT_NONE : (int)pd.mat.type;
        m.is_emissive = pd.mat.is_emissive;
        m.n_textures = pd.mat.specs_taken;
        m.id = pd.mat.id;
        for (int k = 0; k < 16; k++) {
            const TextureDescriptor &t = pd.mat.textures[k];
            PtrTexture &o = m.textures[k];
            o.type = t.type;
            memcpy(o.spec, &t.spec_val, 12);
            o.fval = t.fval;
            o.image = -1;
            if (t.type == 2 && t.image) {
                auto it = imageIdx.find(t.image);
                if (it == imageIdx.end()) {
                    PtrImage im;
                    im.width = t.image->width; im.height = t.image->height; im.rgb = (const float *)t.image->data;
                    it = imageIdx.insert({t.image, (int)h->images.size()}).first;
                    h->images.push_back(im);
                }
                o.image = it->second;
            }
            m.vals[k] = pd.mat.vals[k];
        }
        p.medium = flat_medium(pd.mediumDesc);
        h->prims.push_back(p);
    }
    PtrFlatScene &fs = h->fs;
    memset(&fs, 0, sizeof(fs));
    fs.n_primitives = (int)h->prims.size(); fs.primitives = h->prims.data();
    fs.n_meshes = (int)h->meshes.size(); fs.meshes = h->meshes.data();
    fs.n_clouds = (int)h->clouds.size(); fs.clouds = h->clouds.data();
    fs.n_images = (int)h->images.size(); fs.images = h->images.data();
    fs.camera_medium = flat_medium(g_cameraMedium);
    fs.has_envmap = g_lightMap.valid ? 1 : 0;
    if (g_lightMap.valid) {
        h->lm = g_lightMap;
        PtrEnvMap &e = fs.envmap;
        e.light_to_world = TransformToFlat(h->lm.toWorld);
        e.width = h->lm.width; e.height = h->lm.height; e.nu = h->lm.nu; e.nv = h->lm.nv;
        e.texels = (const float *)h->lm.texels.data();
        e.func = h->lm.func.data(); e.cdf = h->lm.cdf.data(); e.func_int = h->lm.funcInt.data(); e.marg_cdf = h->lm.margCdf.data();
        e.marg_int = h->lm.margInt;
    }
    h->distant = g_distantLights;
    fs.n_distant = (int)h->distant.size();
    fs.distant = h->distant.empty() ? nullptr : h->distant.data();
    return h;
}
const PtrFlatScene *FlatSceneGet(const FlatSceneHolder *h) { return &h->fs; }
void FlatSceneFree(FlatSceneHolder *h) { delete h; }

void PrepareSceneForRendering(Aggregator *scene) {
    if (!scene->ctx) scene->ctx = default_ctx();
    FlatSceneHolder *h = FlattenScene();
    die_on(ptr_scene_upload(scene->ctx, &h->fs), "ptr_scene_upload");
    FlatSceneFree(h);
}

/* ---- render entry points ------------------------------------------------------------------------------- */
Image *CreateImage(int res_x, int res_y) {
    Image *im = new Image();
    im->width = res_x; im->height = res_y; im->pixels_count = res_x * res_y;
    im->pixels = new Pixel[(size_t)im->pixels_count]();
    im->owner = nullptr;
    return im;
}
void ImageFree(Image *image) {
    if (image) { delete[] image->pixels; delete image; }
}
void SetIntegrator(int id) { g_integrator = id; }
void SetupPixels(Image *image, unsigned long long seed) {
    die_on(ptr_film_init(default_ctx(), image->width, image->height, seed), "ptr_film_init");
}
void Render(Image *image, Aggregator *scene, Camera *camera, int ns) {
    image->owner = scene;
    PtrCamera pc = CameraToFlat(*camera);
    die_on(ptr_camera_set(scene->ctx, &pc), "ptr_camera_set");
    die_on(ptr_render(scene->ctx, ns, g_integrator), "ptr_render");
}
void ImageSync(Image *image) {
    PtrContext *ctx = image->owner ? image->owner->ctx : default_ctx();
    size_t n = (size_t)image->pixels_count;
    std::vector<float> we(3 * n);
    std::vector<double> aw(n);
    std::vector<int64_t> sm(n);
    die_on(ptr_film_read(ctx, we.data(), aw.data(), sm.data()), "ptr_film_read");
    for (size_t i = 0; i < n; i++) {
        image->pixels[i].we = Spectrum(we[3 * i], we[3 * i + 1], we[3 * i + 2]);
        image->pixels[i].accWeight = aw[i];
        image->pixels[i].samples = (long)sm[i];
    }
}
