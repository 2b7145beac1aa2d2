/*
 * pt_scenes.cpp — the benchmark / test scenes, written against the kept scene API exactly the
 * way src/main.cpp:459-1095 writes its scene functions.  The author's assets (dragon.obj, EXR
 * sky maps, SPH frames; main.cpp:21-22) are not available, so geometry and the environment map
 * are synthesised in code with fixed seeds; everything else (cameras, materials,
 * transforms) follows the cited reference scene.
 *
 * A small C interface at the bottom lets Python drive these builders and fetch the flat scene.
 */
#include "pt_scene.h"
#include <cstdint>
#include <cstdlib>

namespace {

struct Pcg { /* PCG32 for synthetic data only (never on the render path) */
    uint64_t state, inc;
    explicit Pcg(uint64_t seed) : state(0), inc((seed << 1u) | 1u) { next(); state += 0x853c49e6748fea9bULL; next(); }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    float uniform() { return (next() >> 8) * (1.0f / 16777216.0f); }
};

static std::string g_sky_exr; /* when set (pth_scene_sky_exr), ZooScene reads its sky from this file */

/* analytic sky + sun, equirectangular, stands in for skylight-sunset.exr (main.cpp:479) */
std::vector<Spectrum> MakeSky(int w, int h) {
    std::vector<Spectrum> px((size_t)w * h);
    const double sunTheta = 1.05, sunPhi = 2.2;
    double sx = std::sin(sunTheta) * std::cos(sunPhi), sy = std::sin(sunTheta) * std::sin(sunPhi), sz = std::cos(sunTheta);
    for (int v = 0; v < h; v++) {
        double theta = Pi * (v + 0.5) / h;
        for (int u = 0; u < w; u++) {
            double phi = 2 * Pi * (u + 0.5) / w;
            double dx = std::sin(theta) * std::cos(phi), dy = std::sin(theta) * std::sin(phi), dz = std::cos(theta);
            double up = dz; /* light-space +z is "up" after the scene's RotateX(-95) */
            double t = 0.5 * (up + 1.0);
            double r = (1 - t) * 0.95 + t * 0.35, g = (1 - t) * 0.75 + t * 0.55, b = (1 - t) * 0.55 + t * 0.95;
            if (up < 0) { r *= 0.15; g *= 0.15; b *= 0.15; } /* dark ground half, like the half-black EXRs (light.h:86-92) */
            double c = dx * sx + dy * sy + dz * sz;
            double sun = c > 0 ? std::pow(c, 400.0) * 40.0 + std::pow(c, 12.0) * 0.6 : 0;
            px[(size_t)v * w + u] = Spectrum((Float)(r + sun), (Float)(g + 0.9 * sun), (Float)(b + 0.7 * sun));
        }
    }
    return px;
}

/* closed (2,3) torus-knot tube, nu x nv quads -> 2*nu*nv triangles (660x660 -> 871,200) */
ParsedMesh *MakeKnotMesh(int nu, int nv, uint64_t seed, const Transform &toWorld) {
    std::vector<Point3f> P((size_t)nu * nv);
    std::vector<Point3i> I((size_t)nu * nv * 6);
    Pcg rng(seed);
    const double tube = 0.42;
    for (int i = 0; i < nu; i++) {
        double t = 2 * Pi * i / nu;
        auto curve = [](double s, double *o) {
            double r = 2 + std::cos(3 * s);
            o[0] = r * std::cos(2 * s); o[1] = r * std::sin(2 * s); o[2] = 1.2 * std::sin(3 * s);
        };
        double c[3], c1[3];
        curve(t, c);
        curve(t + 1e-4, c1);
        double T[3] = {c1[0] - c[0], c1[1] - c[1], c1[2] - c[2]};
        double tl = std::sqrt(T[0] * T[0] + T[1] * T[1] + T[2] * T[2]);
        for (double &x : T) x /= tl;
        double N[3] = {-c[0], -c[1], 0}; /* towards the axis, made orthogonal to T */
        double d = N[0] * T[0] + N[1] * T[1] + N[2] * T[2];
        for (int k = 0; k < 3; k++) N[k] -= d * T[k];
        double nl = std::sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
        for (double &x : N) x /= nl;
        double B[3] = {T[1] * N[2] - T[2] * N[1], T[2] * N[0] - T[0] * N[2], T[0] * N[1] - T[1] * N[0]};
        for (int j = 0; j < nv; j++) {
            double a = 2 * Pi * j / nv;
            double rr = tube * (1.0 + 0.04 * std::sin(9 * a + 5 * t)) + 0.004 * (rng.uniform() - 0.5);
            double ca = std::cos(a) * rr, sa = std::sin(a) * rr;
            P[(size_t)i * nv + j] = Point3f((Float)(c[0] + ca * N[0] + sa * B[0]), (Float)(c[1] + ca * N[1] + sa * B[1]),
                                            (Float)(c[2] + ca * N[2] + sa * B[2]));
        }
    }
    size_t k = 0;
    for (int i = 0; i < nu; i++)
        for (int j = 0; j < nv; j++) {
            int i1 = (i + 1) % nu, j1 = (j + 1) % nv;
            int a = i * nv + j, b = i1 * nv + j, c = i1 * nv + j1, d = i * nv + j1;
            I[k++] = Point3i(a, 0, 0); I[k++] = Point3i(b, 0, 0); I[k++] = Point3i(c, 0, 0);
            I[k++] = Point3i(a, 0, 0); I[k++] = Point3i(c, 0, 0); I[k++] = Point3i(d, 0, 0);
        }
    return ParsedMeshFromData(toWorld, nu * nv * 2, I.data(), nu * nv, P.data());
}

/* ---- scenes --------------------------------------------------------------------------------- */

/* c1: geometry of CornellRoomBasic (main.cpp:748-802); variant bit 0 keeps the mirror box and
 * adds the glass material the reference declares; otherwise both boxes are white matte and the
 * textured floor is white (no assets), as SURVEY §8d prescribes. */
void CornellScene(Camera *camera, Float aspect, int variant) {
    camera->Config(Point3f(0, 4, -25), Point3f(0, 4, 0.0f), vec3f(0, 1, 0), 40, aspect);
    MaterialDescriptor white = MakeMatteMaterial(Spectrum(.73, .73, .73));
    MaterialDescriptor green = MakeMatteMaterial(Spectrum(.05, .65, .05));
    MaterialDescriptor red = MakeMatteMaterial(Spectrum(.65, .05, .05));
    MaterialDescriptor mirror = MakeMirrorMaterial(Spectrum(.9));
    MaterialDescriptor glass = MakeGlassMaterial(Spectrum(1), Spectrum(1), 1.34);
    Float height = 14;
    Float width = height * 1.2;
    Float extend = 30;
    InsertPrimitive(MakeRectangle(Translate(0, height / 2, 0) * RotateY(0), width, height), white);
    InsertPrimitive(MakeRectangle(Translate(-width / 2, height / 2, -height / 2) * RotateY(90), width + extend, height), red);
    InsertPrimitive(MakeRectangle(Translate(width / 2, height / 2, -height / 2) * RotateY(90), width + extend, height), green);
    Float bottomExt = 5;
    InsertPrimitive(MakeRectangle(Translate(0, 0, -(height + bottomExt) / 2) * RotateX(90), width, height + bottomExt), white);
    InsertPrimitive(MakeRectangle(Translate(0, height / 2, -28), width, height), white);
    InsertPrimitive(MakeRectangle(Translate(0, height, -height / 2) * RotateX(90), width, height), white);
    /* variant & 4: the boxes in the Disney material (all five lobes on the first, diffuse + retro + clearcoat on the second) */
    MaterialDescriptor disney1 = MakeDisneyMaterial(Spectrum(.7, .35, .2), .6, 1.f, .5, 1.f, .8, .4);
    MaterialDescriptor disney2 = MakeDisneyMaterial(Spectrum(.3, .45, .7), .3, 0.f, .5, .6, .2, 0.f);
    /* variant & 8: the boxes as CornellRoomBasic has them (white, mirror): the scene of the reference's images/cornell.png */
    InsertPrimitive(MakeBox(Translate(-3.5, 2, -width * 0.5) * RotateY(-25), 4, 4, 4), (variant & 4) ? disney1 : (variant & 1) ? glass : white);
    InsertPrimitive(MakeBox(Translate(3.0, 4.25, -width * 0.25) * RotateY(30), 4.5, 8.5, 4.5), (variant & 4) ? disney2 : (variant & 9) ? mirror : white);
    MaterialDescriptor matEm = MakeEmissive(Spectrum(0.992, 0.964, 0.390) * 6);
    InsertPrimitive(MakeDisk(Translate(0, height - 0.01, -height / 2) * RotateX(90), 0, width * 0.15, 0.f, 360), matEm);
    if (variant & 2) InsertCameraMedium(MakeMedium(Spectrum(.008), Spectrum(.01), 0)); /* CornellRoomSkull, main.cpp:851 */
}

/* c2: DragonScene (main.cpp:459-487) with the knot mesh standing in for dragon.obj and the
 * analytic sky for skylight-sunset.exr.  variant: 0 = microfacet glass, 1 = plastic (as shipped),
 * 2 = specular glass, 3 = matte. */
void KnotScene(Camera *camera, Float aspect, int segments, int variant) {
    camera->Config(Point3f(60.f, 69.f, -43.f), Point3f(-12.12833, 0., 3.16229), vec3f(0.f, 1.f, 0.f), 38.f, aspect);
    MaterialDescriptor gray = MakePlasticMaterial(Spectrum(.1, .1, .1), Spectrum(.7, .7, .7), 0.1);
    MaterialDescriptor blue = MakePlasticMaterial(Spectrum(.1f, .1f, .4f), Spectrum(0.6f), 0.03);
    MaterialDescriptor mglass = MakeGlassMaterial(Spectrum(1), Spectrum(1), 1.5, .05, .05);
    MaterialDescriptor sglass = MakeGlassMaterial(Spectrum(1), Spectrum(1), 1.5);
    MaterialDescriptor matte = MakeMatteMaterial(Spectrum(.6, .5, .4));
    SphereDescriptor bottomWall = MakeSphere(Translate(0, -1500, 0), 1500);
    static std::vector<Spectrum> sky;
    if (sky.empty()) sky = MakeSky(512, 256);
    InsertLightMap(sky.data(), 512, 256, RotateX(-95) * RotateZ(50), Spectrum(2.5));
    Transform toWorld = Translate(-12, 18, 3) * Scale(7.5) * RotateZ(-1) * RotateY(70) * RotateX(90);
    ParsedMesh *mesh = MakeKnotMesh(segments, segments, 7, toWorld);
    MeshDescriptor knot = MakeMesh(mesh);
    InsertPrimitive(knot, variant == 0 ? mglass : (variant == 1 ? blue : (variant == 2 ? sglass : matte)));
    InsertPrimitive(bottomWall, gray);
}

/* White furnace: one convex object inside a constant environment of radiance 1.  Nothing from the reference -
 * a scene whose answer is known in closed form (tests/test_*_furnace.py): a convex Lambertian body shows its
 * albedo, a mirror its reflectance, a non-absorbing dielectric exactly 1.
 * variant: 0 matte 0.5 | 1 mirror 0.9 | 2 specular glass R = T = 1 | 3 rough glass | 4 matte box instead of a sphere |
 * 5 a ball of purely scattering medium (sigma_a = 0: also exactly 1) | 6 matte octahedron as a triangle mesh |
 * 7 a matte floor under a sky that is 1 above the horizon and 0 below (irradiance pi -> the floor shows its albedo;
 *   the light map is anything but uniform, so this goes through the Distribution2D importance sampling) */
void FurnaceScene(Camera *camera, Float aspect, int variant) {
    camera->Config(Point3f(0.f, 0.f, -6.f), Point3f(0.f, 0.f, 0.f), vec3f(0.f, 1.f, 0.f), 30.f, aspect);
    if (variant == 7) {
        camera->Config(Point3f(0.f, 4.f, -6.f), Point3f(0.f, 0.f, 0.f), vec3f(0.f, 1.f, 0.f), 20.f, aspect);
        static std::vector<Spectrum> half(64 * 32);
        for (int v = 0; v < 32; v++)
            for (int u = 0; u < 64; u++) half[(size_t)v * 64 + u] = Spectrum(v < 16 ? 1.f : 0.f); /* theta < pi/2: the light's +z side */
        InsertLightMap(half.data(), 64, 32, RotateX(-90), Spectrum(1.0));
        InsertPrimitive(MakeRectangle(RotateX(90), 200, 200), MakeMatteMaterial(Spectrum(.5, .5, .5)));
        return;
    }
    static std::vector<Spectrum> sky(16 * 8, Spectrum(1.f));
    InsertLightMap(sky.data(), 16, 8, RotateX(-90), Spectrum(1.0));
    MaterialDescriptor mat = variant == 1 ? MakeMirrorMaterial(Spectrum(.9))
                           : variant == 2 ? MakeGlassMaterial(Spectrum(1), Spectrum(1), 1.5)
                           : variant == 3 ? MakeGlassMaterial(Spectrum(1), Spectrum(1), 1.5, .1, .1)
                           : variant == 8 ? MakeDisneyMaterial(Spectrum(.5, .5, .5), .5)                       /* diffuse + retro */
                           : variant == 9 ? MakeDisneyMaterial(Spectrum(.5, .5, .5), .5, 1.f, .5, 1.f, .8, .4)  /* all five lobes */
                                          : MakeMatteMaterial(Spectrum(.5, .5, .5));
    if (variant == 4) InsertPrimitive(MakeBox(RotateY(30) * RotateX(20), 1.4, 1.4, 1.4), mat);
    else if (variant == 5) InsertPrimitive(MakeSphere(Translate(0, 0, 0), 1), MakeMedium(Spectrum(0.f), Spectrum(.8f), 0.3f));
    else if (variant == 6) {
        static Point3f P[6] = {Point3f(1, 0, 0), Point3f(-1, 0, 0), Point3f(0, 1, 0), Point3f(0, -1, 0), Point3f(0, 0, 1), Point3f(0, 0, -1)};
        static Point3i I[24];
        const int f[8][3] = {{0, 2, 4}, {2, 1, 4}, {1, 3, 4}, {3, 0, 4}, {2, 0, 5}, {1, 2, 5}, {3, 1, 5}, {0, 3, 5}};
        for (int t = 0; t < 8; t++)
            for (int c = 0; c < 3; c++) I[3 * t + c] = Point3i(f[t][c], -1, -1);
        InsertPrimitive(MakeMesh(ParsedMeshFromData(RotateY(25) * RotateX(15), 8, I, 6, P)), mat);
    } else InsertPrimitive(MakeSphere(Translate(0, 0, 0), 1), mat);
}

/* material zoo on spheres/disks/rects under an area light + sky: exercises every BxDF the
 * materials instantiate (BoxesScene-like, main.cpp:489-570, small) */
void ZooScene(Camera *camera, Float aspect, int variant) {
    camera->Config(Point3f(0, 9, -26), Point3f(0, 3, 0), vec3f(0, 1, 0), 40, aspect);
    MaterialDescriptor mats[] = {
        MakeMatteMaterial(Spectrum(.7, .3, .2), 0),
        MakeMatteMaterial(Spectrum(.3, .6, .3), 0.4),
        MakeMirrorMaterial(Spectrum(.9)),
        MakeGlassMaterial(Spectrum(1), Spectrum(1), 1.5),
        MakeGlassMaterial(Spectrum(.9), Spectrum(.95), 1.5, .1, .2),
        MakeMetalMaterial(Spectrum(.2, .92, 1.1), Spectrum(3.9, 2.45, 2.14), .05, .05),
        MakePlasticMaterial(Spectrum(.2, .2, .7), Spectrum(.6), 0.1),
        MakeUberMaterial(Spectrum(.25, .5, .25), Spectrum(.3), Spectrum(.1), Spectrum(.2), .1, .1, Spectrum(.9), 1.4),
        MakeTranslucentMaterial(Spectrum(.5, .4, .3), Spectrum(.3), Spectrum(.6), Spectrum(.4), .2),
        MakePlayGroundMaterial(Spectrum(.85, .42, .52)),
    };
    int n = sizeof(mats) / sizeof(mats[0]);
    for (int i = 0; i < n; i++) {
        Float x = -9.f + 4.5f * (i % 5), z = (i / 5) * 5.f - 2.f;
        if (i % 3 == 2) InsertPrimitive(MakeBox(Translate(x, 1.5, z) * RotateY(20 + 10 * i), 2.6, 3, 2.6), mats[i]);
        else InsertPrimitive(MakeSphere(Translate(x, 1.6, z), 1.6), mats[i]);
    }
    InsertPrimitive(MakeRectangle(Translate(0, 0, 0) * RotateX(90), 60, 60), MakeMatteMaterial(Spectrum(.6)));
    InsertPrimitive(MakeRectangle(Translate(0, 8, 12), 60, 30), MakePlasticMaterial(Spectrum(.4, .4, .45), Spectrum(.3), 0.2));
    InsertPrimitive(MakeDisk(Translate(0, 14, 0) * RotateX(90), 0, 4, 1, 300), MakeEmissive(Spectrum(9, 8.5, 7)));
    InsertPrimitive(MakeSphere(Translate(8, 9, -4), 1.0), MakeEmissive(Spectrum(12, 10, 6)));
    if (variant & 1) {
        static std::vector<Spectrum> sky;
        if (sky.empty()) sky = MakeSky(256, 128);
        if (!g_sky_exr.empty()) InsertEXRLightMap(g_sky_exr.c_str(), RotateX(-90) * RotateZ(90), Spectrum(1.0)); /* the file route */
        else InsertLightMap(sky.data(), 256, 128, RotateX(-90) * RotateZ(90), Spectrum(1.0));
    }
    if (variant & 2) InsertCameraMedium(MakeMedium(Spectrum(.004), Spectrum(.012), 0.3));
}

/* OrigamiScene (main.cpp:854-911): the four signed-distance toys with per-face PlayGround colours
 * over a matte ground, under the generated sky (the reference loads skylight-sunset.exr) */
void OrigamiScene(Camera *camera, Float aspect) {
    camera->Config(Point3f(0.f, 62.f, 70.f), Point3f(-10, 23, 0), vec3f(0.f, 1.f, 0.f), 50.f, aspect);
    std::vector<Spectrum> boatSpectrums({SpectrumFromURGB(139, 69, 19), Spectrum(.87, .12, .27), SpectrumFromURGB(244, 164, 96)});
    std::vector<Spectrum> whaleSpectrums({Spectrum(.02, .21, .54), Spectrum(.1, .32, .76), Spectrum(.86, .86, .98)});
    MaterialDescriptor gray = MakeMatteMaterial(Spectrum(0.9, 0.9, 0.7));
    MaterialDescriptor dragonMat = MakePlayGroundMaterial(Spectrum(.85, .12, .02));
    MaterialDescriptor boatMat = MakePlayGroundMaterial(boatSpectrums);
    MaterialDescriptor whaleMat = MakePlayGroundMaterial(whaleSpectrums);
    MaterialDescriptor birdMat = MakePlayGroundMaterial(Spectrum(.85, .42, .52));
    InsertPrimitive(MakeRectangle(Translate(0, 0, 0) * RotateX(90), 2000, 2000), gray);
    static std::vector<Spectrum> sky;
    if (sky.empty()) sky = MakeSky(256, 128);
    InsertLightMap(sky.data(), 256, 128, RotateX(-90) * RotateZ(90), Spectrum(2.5));
    Float len = 15;
    Float h = 1.5;
    Float h2 = h * 0.5 * len;
    Transform transf = Translate(-len * 0.5, h2 * 0.93, 0) * RotateX(-20);
    Float depth = 0.5;
    Bounds3f bound(Point3f(-len, -h2, -depth), Point3f(len, h2, depth));
    ProcToyDescriptor dragon = MakeProceduralToy(transf, bound, 1);
    transf = Translate(-len * 2.5, h2 * 0.93, len);
    ProcToyDescriptor whale = MakeProceduralToy(transf, bound, 2);
    len = 19;
    h = 2.0;
    h2 = h * 0.5 * len;
    transf = Translate(-len * 0.5, h2, -len) * RotateY(60);
    bound = Bounds3f(Point3f(-len, -h2, -len), Point3f(len, h2, len));
    ProcToyDescriptor boat = MakeProceduralToy(transf, bound, 0);
    InsertPrimitive(dragon, dragonMat);
    InsertPrimitive(boat, boatMat);
    InsertPrimitive(whale, whaleMat);
    Bounds3f bo2(Point3f(-len, -len, -len), Point3f(len, len, len));
    transf = Translate(-3 * len, 2.0 * len, -len) * RotateY(-45);
    InsertPrimitive(MakeProceduralToy(transf, bo2, 3), birdMat);
    transf = Translate(len, 1.6 * len, -len * 0.5) * RotateY(200);
    InsertPrimitive(MakeProceduralToy(transf, bo2, 3), dragonMat);
}

/* c5 (volume half): VolumetricCausticsScene verbatim, main.cpp:987-1021 */
void VolumetricCausticsScene(Camera *camera, Float aspect) {
    camera->Config(Point3f(0.f, 18.f, 70.f), Point3f(0, 21.15341, 0), vec3f(0.f, 1.f, 0.f), 40.f, aspect);
    MaterialDescriptor yellow = MakeMatteMaterial(Spectrum(.7, .7, .0));
    MaterialDescriptor greenGlass = MakeGlassMaterial(Spectrum(1), Spectrum(1), 1.5);
    Transform er = Translate(0, -1.29, 0) * RotateX(90);
    RectDescriptor bottomWall = MakeRectangle(er, 1000, 1000);
    InsertPrimitive(bottomWall, yellow);
    MaterialDescriptor matEm = MakeEmissive(Spectrum(0.992, 0.964, 0.890) * 35);
    SphereDescriptor lightSphere = MakeSphere(Translate(15, 60, 3), 4);
    SphereDescriptor lightSphere2 = MakeSphere(Translate(-15, 60, 3), 4);
    InsertPrimitive(lightSphere, matEm);
    InsertPrimitive(lightSphere2, matEm);
    MediumDescriptor medium = MakeMedium(Spectrum(.0007), Spectrum(0.005), 0);
    Transform glassT = Translate(0, 26, 0) * RotateY(45);
    SphereDescriptor spehre = MakeSphere(glassT, 13);
    InsertPrimitive(spehre, greenGlass);
    InsertCameraMedium(medium);
}

/* c4 stand-in: FluidDragonInBox (main.cpp:1023-1095) with a synthetic dam-break column:
 * jittered lattice in an L-shaped region of the 2.25 x 2.85 x 4.9 box, radius 0.012, split in
 * six colour classes by height. */
static std::vector<std::vector<vec3f>> g_cloudStore;
void FluidScene(Camera *camera, Float aspect, int nParticles) {
    camera->Config(Point3f(0.f, 1.5f, 7.0f), Point3f(0.f, -0.5f, 0.f), vec3f(0.f, 1.f, 0.f), 35.f, aspect);
    MaterialDescriptor white = MakeMatteMaterial(Spectrum(2.0));
    Float radius = 0.012;
    MaterialDescriptor materials[] = {
        MakeMatteMaterial(Spectrum(0.97, 0.00, 0.10)), MakeMatteMaterial(Spectrum(0.90, 0.44, 0.10)),
        MakeMatteMaterial(Spectrum(0.95, 0.76, 0.30)), MakeMatteMaterial(Spectrum(0.45, 0.70, 0.84)),
        MakeMatteMaterial(Spectrum(0.15, 0.40, 0.74)), MakeMatteMaterial(Spectrum(0.78, 0.78, 0.74))};
    g_cloudStore.assign(6, std::vector<vec3f>());
    Pcg rng(11);
    const float bx = 2.25052f * 0.5f, by = 2.85056f * 0.5f, bz = 4.90718f * 0.5f;
    /* L-shape: a tall column near -z plus a shallow pool over the rest */
    double vol = (2 * bx) * (2 * by) * (0.35 * 2 * bz) + (2 * bx) * (0.25 * 2 * by) * (0.65 * 2 * bz);
    double spacing = std::cbrt(vol / (nParticles > 0 ? nParticles : 1));
    int made = 0;
    for (double y = -by + radius; y < by && made < nParticles; y += spacing)
        for (double z = -bz + radius; z < bz && made < nParticles; z += spacing) {
            bool column = z < -bz + 0.35 * 2 * bz;
            if (!column && y > -by + 0.25 * 2 * by) continue;
            for (double x = -bx + radius; x < bx && made < nParticles; x += spacing) {
                vec3f p((Float)(x + 0.3 * spacing * (rng.uniform() - .5f)), (Float)(y + 0.3 * spacing * (rng.uniform() - .5f)),
                        (Float)(z + 0.3 * spacing * (rng.uniform() - .5f)));
                int level = (int)(6 * (y + by) / (2 * by));
                if (level > 5) level = 5;
                g_cloudStore[level].push_back(p);
                made++;
            }
        }
    for (int l = 0; l < 6; l++) {
        if (g_cloudStore[l].empty()) continue;
        ParticleCloudDescripor desc = MakeParticleCloud(g_cloudStore[l].data(), (int)g_cloudStore[l].size(), radius);
        InsertPrimitive(desc, materials[l]);
    }
    static std::vector<Spectrum> sky;
    if (sky.empty()) sky = MakeSky(512, 256);
    InsertLightMap(sky.data(), 512, 256, RotateX(-90) * RotateZ(90), Spectrum(2.5));
    InsertPrimitive(MakeRectangle(Translate(0, -1.43, 0) * RotateX(90), 100, 100), white);
    InsertPrimitive(MakeRectangle(Translate(0, 0, -20), 100, 100), white);
}

/* c5 (subsurface half): SubsurfaceSpheres (main.cpp:914-952) + a KdSubsurface knot */
void SubsurfaceScene(Camera *camera, Float aspect, int segments) {
    camera->Config(Point3f(0.f, 62.f, 90.f), Point3f(-10, 0, -20), vec3f(0.f, 1.f, 0.f), 42.f, aspect);
    MaterialDescriptor gray = MakePlasticMaterial(Spectrum(.1), Spectrum(.7), 0.1);
    InsertPrimitive(MakeRectangle(Translate(0, 0, 0) * RotateX(90), 1000, 1000), gray);
    MaterialDescriptor ket = MakeSubsurfaceMaterial("Ketchup", 1, 1.33, 0, 0.05, 0.05);
    MaterialDescriptor spec = MakeKdSubsurfaceMaterial(Spectrum(.2, .5, .7), Spectrum(1), Spectrum(1), Spectrum(.1), 0.05, 0.05, 1.33, 1);
    MaterialDescriptor kd0 = MakeKdSubsurfaceMaterial(Spectrum(.2, .5, .7), Spectrum(1), Spectrum(1), Spectrum(.1), 0, 0, 1.33, 1);
    MaterialDescriptor apple = MakeSubsurfaceMaterial("Apple", 1, 1.33, 0, 0.05, 0.05);
    static std::vector<Spectrum> sky;
    if (sky.empty()) sky = MakeSky(512, 256);
    InsertLightMap(sky.data(), 512, 256, RotateX(-90) * RotateZ(90), Spectrum(2.5));
    InsertPrimitive(MakeSphere(Translate(20, 20, 10), 18), ket);
    InsertPrimitive(MakeSphere(Translate(-30, 20, 10), 18), apple);
    InsertPrimitive(MakeSphere(Translate(6, 20, -20), 18), spec);
    if (segments > 0) {
        Transform toWorld = Translate(-60, 22, -20) * Scale(6.0) * RotateX(90);
        InsertPrimitive(MakeMesh(MakeKnotMesh(segments, segments, 9, toWorld)), kd0);
    } else InsertPrimitive(MakeSphere(Translate(-60, 20, -20), 18), kd0);
}


/* c3: an indoor room in the manner of CornellRoom (main.cpp:626-711: image-textured rectangle walls, several meshes that
 * came with their own materials, an emissive rectangle under the ceiling) at the size SURVEY 8(d) gives configs[2]:
 * 40 rectangle / SDF-box primitives (walls, furniture), 8 knot meshes of 2 * segments^2 triangles each (224 -> 100,352)
 * standing in for the afrodite / shield OBJ files, one material class each, and 64 small emitters (a ceiling grid of 48
 * rectangles and 16 sphere bulbs along the walls; the light table holds 256, primitive.h:93).  The wall textures are
 * generated (the reference loads greek.jpg / orange.png).  No sky: every path stays inside. */
static std::vector<Spectrum> g_roomTex[2];
static ImageData g_roomImg[2];
void RoomScene(Camera *camera, Float aspect, int segments) {
    camera->Config(Point3f(-1, 7, -29), Point3f(0, 6.0f, 0.0f), vec3f(0, 1, 0), 48, aspect);
    const Float height = 17, width = 26, zback = 15, zfront = -32;
    const Float depth = zback - zfront, zmid = 0.5f * (zback + zfront);
    /* two generated wall textures: a two-tone checker with a frieze band, and a brick pattern */
    for (int k = 0; k < 2; k++) {
        const int tw = 256, th = 128;
        g_roomTex[k].resize((size_t)tw * th);
        for (int y = 0; y < th; y++)
            for (int x = 0; x < tw; x++) {
                Spectrum c;
                if (k == 0) {
                    bool chk = ((x / 16) + (y / 16)) & 1;
                    c = chk ? Spectrum(.80, .74, .62) : Spectrum(.55, .42, .30);
                    if (y > 96 && y < 112) c = Spectrum(.25, .35, .60);
                } else {
                    int row = y / 12, off = (row & 1) ? 12 : 0;
                    bool mortar = (y % 12) == 0 || ((x + off) % 24) == 0;
                    c = mortar ? Spectrum(.75, .73, .70) : Spectrum(.62 + .05f * ((row * 7 + (x + off) / 24 * 3) % 4), .30, .22);
                }
                g_roomTex[k][(size_t)y * tw + x] = c;
            }
        g_roomImg[k].data = g_roomTex[k].data();
        g_roomImg[k].width = tw;
        g_roomImg[k].height = th;
        g_roomImg[k].is_valid = 1;
    }
    MaterialDescriptor white = MakeMatteMaterial(Spectrum(.73, .73, .73));
    MaterialDescriptor orange = MakeMatteMaterial(Spectrum(.73, .53, .33));
    MaterialDescriptor wood = MakeMatteMaterial(Spectrum(.45, .30, .18), 0.3);
    MaterialDescriptor dark = MakePlasticMaterial(Spectrum(.08, .08, .09), Spectrum(.5), 0.05);
    MaterialDescriptor checker = MakeMatteMaterial(MakeTexture(&g_roomImg[0]));
    MaterialDescriptor brick = MakeMatteMaterial(MakeTexture(&g_roomImg[1]));
    MaterialDescriptor mirror = MakeMirrorMaterial(Spectrum(.9));
    MaterialDescriptor glass = MakeGlassMaterial(Spectrum(1), Spectrum(1), 1.5);
    MaterialDescriptor floorMat = MakeUberMaterial(Spectrum(.35, .33, .30), Spectrum(.25), Spectrum(0), Spectrum(0), 0.05, 0.05, Spectrum(1), 1.5f);
    int nprim = 0;
    /* 6 walls */
    InsertPrimitive(MakeRectangle(Translate(0, height / 2, zback), width, height), checker); nprim++;
    InsertPrimitive(MakeRectangle(Translate(-width / 2, height / 2, zmid) * RotateY(90), depth, height), brick); nprim++;
    InsertPrimitive(MakeRectangle(Translate(width / 2, height / 2, zmid) * RotateY(90), depth, height), brick); nprim++;
    InsertPrimitive(MakeRectangle(Translate(0, 0, zmid) * RotateX(90), width, depth), floorMat); nprim++;
    InsertPrimitive(MakeRectangle(Translate(0, height, zmid) * RotateX(90), width, depth), white); nprim++;
    InsertPrimitive(MakeRectangle(Translate(0, height / 2, zfront), width, height), orange); nprim++;
    /* furniture: a long table (top + 4 legs), two benches (top + 2 legs each), four pedestals, two shelves of three boards, a
     * wall mirror, two picture rectangles, a glass display box, a rug, four pillars, a door panel: 34 primitives */
    InsertPrimitive(MakeBox(Translate(0, 3.0, 2), 12, 0.4, 5), wood); nprim++;
    for (int i = 0; i < 4; i++) { InsertPrimitive(MakeBox(Translate((i & 1) ? 5.5 : -5.5, 1.4, (i & 2) ? 4.0 : 0.0), 0.5, 2.8, 0.5), wood); nprim++; }
    for (int b = 0; b < 2; b++) {
        Float x = b ? 9.5 : -9.5;
        InsertPrimitive(MakeBox(Translate(x, 1.6, -8) * RotateY(b ? 12 : -12), 3, 0.3, 7), dark); nprim++;
        InsertPrimitive(MakeBox(Translate(x, 0.75, -10.5), 2.4, 1.5, 0.4), dark); nprim++;
        InsertPrimitive(MakeBox(Translate(x, 0.75, -5.5), 2.4, 1.5, 0.4), dark); nprim++;
    }
    for (int i = 0; i < 4; i++) { InsertPrimitive(MakeBox(Translate(-9 + 6 * i, 1.0, 10.5) * RotateY(15 * i), 2.4, 2.0, 2.4), white); nprim++; }
    for (int sft = 0; sft < 2; sft++)
        for (int i = 0; i < 3; i++) { InsertPrimitive(MakeBox(Translate(sft ? 12.4 : -12.4, 5 + 3 * i, 6), 1.0, 0.25, 8), wood); nprim++; }
    InsertPrimitive(MakeRectangle(Translate(-width / 2 + 0.02, 8, -12) * RotateY(90), 7, 9), mirror); nprim++;
    InsertPrimitive(MakeRectangle(Translate(-5, 11, zback - 0.02), 6, 4), MakeMatteMaterial(Spectrum(.15, .35, .25))); nprim++;
    InsertPrimitive(MakeRectangle(Translate(6, 11, zback - 0.02), 5, 5), MakeMatteMaterial(Spectrum(.55, .15, .12))); nprim++;
    InsertPrimitive(MakeBox(Translate(0, 4.45, 2) * RotateY(20), 2.5, 2.5, 2.5), glass); nprim++;
    InsertPrimitive(MakeRectangle(Translate(0, 0.01, -10) * RotateX(90), 9, 12), MakeMatteMaterial(Spectrum(.45, .12, .12), 0.5)); nprim++;
    for (int i = 0; i < 4; i++) { InsertPrimitive(MakeBox(Translate((i & 1) ? 11.5 : -11.5, height / 2, (i & 2) ? 13.5 : -20.0), 1.5, height, 1.5), white); nprim++; }
    InsertPrimitive(MakeBox(Translate(8, 4.5, zfront + 0.3), 4, 9, 0.3), wood); nprim++;
    InsertPrimitive(MakeBox(Translate(-8.5, 1.5, -24) * RotateY(8), 5, 3, 1.8), wood); nprim++;   /* a low cabinet */
    InsertPrimitive(MakeBox(Translate(-3, 0.9, -4) * RotateY(30), 1.4, 1.8, 1.4), dark); nprim++; /* two stools */
    InsertPrimitive(MakeBox(Translate(3.2, 0.9, -3) * RotateY(-20), 1.4, 1.8, 1.4), dark); nprim++;
    (void)nprim; /* 40 */
    /* 8 meshes, one material class each (CornellRoom gets its materials from the .mtl files) */
    MaterialDescriptor meshMats[8] = {
        MakeMatteMaterial(Spectrum(.70, .68, .62), 0.2),
        MakePlasticMaterial(Spectrum(.15, .25, .55), Spectrum(.5), 0.05),
        MakeMetalMaterial(Spectrum(.2, .92, 1.1), Spectrum(3.9, 2.45, 2.14), .05, .05),
        MakeGlassMaterial(Spectrum(1), Spectrum(1), 1.5, .05, .05),
        MakeUberMaterial(Spectrum(.05), Spectrum(.8), Spectrum(0), Spectrum(0), 0.001, 0.001, Spectrum(1), 1.5f), /* main.cpp:636 */
        MakeTranslucentMaterial(Spectrum(.5, .4, .3), Spectrum(.3), Spectrum(.6), Spectrum(.4), .2),
        MakeMirrorMaterial(Spectrum(.85, .8, .6)),
        MakeMatteMaterial(MakeTexture(&g_roomImg[0])),
    };
    for (int m = 0; m < 8; m++) {
        Transform T;
        if (m < 4) T = Translate(-9 + 6 * m, 3.6, 10.5) * Scale(0.42) * RotateY(40 * m) * RotateX(90); /* on the pedestals */
        else if (m < 6) T = Translate(m == 4 ? -3.5 : 3.5, 4.4, 2) * Scale(0.36) * RotateY(25 * m) * RotateX(70); /* on the table */
        else T = Translate(m == 6 ? -6 : 6, 2.2, -14) * Scale(0.62) * RotateY(-30 * m) * RotateX(90); /* on the floor */
        InsertPrimitive(MakeMesh(MakeKnotMesh(segments, segments, 20 + m, T)), meshMats[m]);
    }
    /* 64 emitters: 8 x 6 downlights just under the ceiling, 16 bulbs along the side walls */
    MaterialDescriptor down = MakeEmissive(Spectrum(0.992, 0.964, 0.790) * 14);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 6; j++)
            InsertPrimitive(MakeRectangle(Translate(-10.5 + 3.0 * i, height - 0.01, zfront + 4 + (depth - 8) * j / 5.0) * RotateX(90), 0.9, 0.9), down);
    MaterialDescriptor bulb = MakeEmissive(Spectrum(1.0, 0.75, 0.45) * 25);
    for (int i = 0; i < 16; i++)
        InsertPrimitive(MakeSphere(Translate((i & 1) ? 12.3 : -12.3, 12.5, zfront + 3 + (depth - 6) * (i / 2) / 7.0), 0.28), bulb);
}

struct BuiltScene {
    FlatSceneHolder *flat;
    Camera camera;
    int rand_calls; /* host rand() calls the scene function made before the BVH build */
};

} // namespace

extern "C" {

/* The light-map tables of arbitrary pixels through InsertLightMap, on the host (on_device = 0) or the device (1): what the
 * tests compare.  Returns the flattened scene that holds them (an empty scene with a sky); pth_scene_flat / pth_scene_free. */
void *pth_lightmap_scene(const float *rgb, int width, int height, const float *scale3, int on_device) {
    BuiltScene *b = new BuiltScene();
    b->rand_calls = 0;
    Aggregator dummy;
    dummy.ctx = (PtrContext *)1;
    BeginScene(&dummy);
    b->camera.Config(Point3f(0, 0, 5), Point3f(0, 0, 0), vec3f(0, 1, 0), 40, 1);
    SetLightMapBuilder(on_device);
    InsertLightMap(reinterpret_cast<const Spectrum *>(rgb), width, height, Translate(0, 0, 0), Spectrum(scale3[0], scale3[1], scale3[2]));
    SetLightMapBuilder(0);
    b->flat = FlattenScene();
    return b;
}

void SetLightMapBuilderC(int on_device) { SetLightMapBuilder(on_device); }

/* name: "cornell" | "knot" | "zoo" | "volcaustic" | "fluid" | "subsurface" | "origami" | "furnace" | "room"; p0/p1: size and variant */
void *pth_scene_build(const char *name, int width, int height, int p0, int p1) {
    std::string n(name);
    BuiltScene *b = new BuiltScene();
    b->rand_calls = 0;
    Float aspect = (Float)width / (Float)height;
    Aggregator dummy;
    dummy.ctx = (PtrContext *)1; /* do not create a GPU context just to describe a scene */
    BeginScene(&dummy);
    if (n == "cornell") CornellScene(&b->camera, aspect, p1);
    else if (n == "knot") KnotScene(&b->camera, aspect, p0 > 0 ? p0 : 660, p1);
    else if (n == "zoo") ZooScene(&b->camera, aspect, p1);
    else if (n == "furnace") FurnaceScene(&b->camera, aspect, p1);
    else if (n == "volcaustic") VolumetricCausticsScene(&b->camera, aspect);
    else if (n == "fluid") FluidScene(&b->camera, aspect, p0 > 0 ? p0 : 4000000);
    else if (n == "subsurface") SubsurfaceScene(&b->camera, aspect, p0);
    else if (n == "origami") OrigamiScene(&b->camera, aspect);
    else if (n == "room") RoomScene(&b->camera, aspect, p0 > 0 ? p0 : 224);
    else { delete b; return nullptr; }
    b->flat = FlattenScene();
    return b;
}
const PtrFlatScene *pth_scene_flat(void *h) { return FlatSceneGet(((BuiltScene *)h)->flat); }
void pth_scene_camera(void *h, PtrCamera *out) { *out = CameraToFlat(((BuiltScene *)h)->camera); }
int pth_scene_rand_calls(void *h) { return ((BuiltScene *)h)->rand_calls; }
void pth_scene_free(void *h) {
    BuiltScene *b = (BuiltScene *)h;
    FlatSceneFree(b->flat);
    delete b;
}

/* the sky the scene functions generate, and a switch that makes ZooScene load its sky from an .exr instead */
void pth_make_sky(int w, int h, float *rgb) {
    std::vector<Spectrum> s = MakeSky(w, h);
    memcpy(rgb, s.data(), s.size() * sizeof(Spectrum));
}
void pth_scene_sky_exr(const char *path) { g_sky_exr = path ? path : ""; }

/* transform factories exposed for the boundary unit tests */
void pth_transform(const char *kind, float a, float b, float c, float d, PtrTransform *out) {
    std::string k(kind);
    Transform t;
    if (k == "translate") t = Translate(a, b, c);
    else if (k == "scale") t = Scale(a, b, c);
    else if (k == "rotx") t = RotateX(a);
    else if (k == "roty") t = RotateY(a);
    else if (k == "rotz") t = RotateZ(a);
    else if (k == "rotate") t = Rotate(a, vec3f(b, c, d));
    else if (k == "trs") t = Translate(a, b, c) * RotateY(d) * Scale(2, 3, 4);
    *out = TransformToFlat(t);
}
void pth_camera_config(const float *eye, const float *at, const float *up, float fov, float aspect, float aperture, PtrCamera *out) {
    Camera c;
    if (aperture > 0) c.Config(Point3f(eye[0], eye[1], eye[2]), Point3f(at[0], at[1], at[2]), vec3f(up[0], up[1], up[2]), fov, aspect, aperture);
    else c.Config(Point3f(eye[0], eye[1], eye[2]), Point3f(at[0], at[1], at[2]), vec3f(up[0], up[1], up[2]), fov, aspect);
    *out = CameraToFlat(c);
}
}
