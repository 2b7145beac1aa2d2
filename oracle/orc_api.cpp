/*
 * orc_api.cpp — ORACLE (test infrastructure only).  PARITY UNPINNED.
 * C interface of the CPU restatement, consumed by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py ONLY.  Builds the reference's object graph from the same flat
 * scene the product's C-ABI takes (include/ptr_api.h), restating MakeSceneGPU / Wrap /
 * MakeLights (src/core/scene.cpp:729-958, src/core/primitive.cpp:164-244,620-642,
 * src/shapes/mesh.cpp:28-69,609-671, src/shapes/particlecloud.cpp:284-337).
 */
#include "../include/ptr_api.h"
#include "orc_scene.h"
#include <atomic>
#include <cstdio>
#include <thread>

namespace orc {
thread_local Counters g_ctr;
}
using namespace orc;

struct OrcScene {
    Scene sc;
    Counters total;
};

static void add_counters(Counters &a, const Counters &b) {
    const uint64_t *s = (const uint64_t *)&b;
    uint64_t *d = (uint64_t *)&a;
    for (size_t i = 0; i < sizeof(Counters) / sizeof(uint64_t); i++) d[i] += s[i];
}

static Tex conv_tex(const PtrTexture &t) {
    Tex o;
    o.type = t.type;
    o.spec = V3(t.spec[0], t.spec[1], t.spec[2]);
    o.fval = t.fval;
    o.image = t.image;
    return o;
}

extern "C" {

void *orc_scene_create(const PtrFlatScene *fs, int rand_calls_before, int meshfix) {
    OrcScene *os = new OrcScene();
    memset(&os->total, 0, sizeof(Counters));
    Scene &sc = os->sc;
    sc.meshfix = meshfix;
    sc.viewMedium = -1;
    GlibcRand rng(1);
    for (int i = 0; i < rand_calls_before; i++) rng.next();

    for (int i = 0; i < fs->n_images; i++) {
        ImageTex im;
        im.width = fs->images[i].width;
        im.height = fs->images[i].height;
        im.data.resize((size_t)im.width * im.height);
        for (size_t k = 0; k < im.data.size(); k++)
            im.data[k] = V3(fs->images[i].rgb[3 * k], fs->images[i].rgb[3 * k + 1], fs->images[i].rgb[3 * k + 2]);
        sc.images.push_back(im);
    }
    /* meshes: Mesh ctor moves vertices to world space (mesh.cpp:61) */
    sc.meshes.resize(fs->n_meshes);
    for (int i = 0; i < fs->n_meshes; i++) {
        const PtrMesh &pm = fs->meshes[i];
        MeshData &m = sc.meshes[i];
        Xf tw = XfFromFlat(pm.to_world.m, pm.to_world.mInv);
        m.nTriangles = pm.n_triangles;
        m.nVertices = pm.n_vertices;
        m.nUvs = pm.n_uvs;
        m.p.resize(pm.n_vertices);
        for (int v = 0; v < pm.n_vertices; v++) m.p[v] = xf_point(tw, V3(pm.p[3 * v], pm.p[3 * v + 1], pm.p[3 * v + 2]));
        m.indices.assign(pm.indices, pm.indices + 9 * (size_t)pm.n_triangles);
        m.has_uv = pm.uv != nullptr;
        if (m.has_uv) {
            m.uv.resize(pm.n_uvs);
            for (int v = 0; v < pm.n_uvs; v++) m.uv[v] = V2(pm.uv[2 * v], pm.uv[2 * v + 1]);
        }
    }
    sc.clouds.resize(fs->n_clouds);
    for (int i = 0; i < fs->n_clouds; i++) {
        CloudData &c = sc.clouds[i];
        c.nParticles = fs->clouds[i].n_particles;
        c.radius = fs->clouds[i].radius;
        c.positions.resize(c.nParticles);
        for (int k = 0; k < c.nParticles; k++)
            c.positions[k] = V3(fs->clouds[i].positions[3 * k], fs->clouds[i].positions[3 * k + 1], fs->clouds[i].positions[3 * k + 2]);
    }

    /* MakeSceneGPU, scene.cpp:891-946 */
    if (fs->camera_medium.is_valid) {
        Medium m;
        m.sigma_a = V3(fs->camera_medium.sigma_a[0], fs->camera_medium.sigma_a[1], fs->camera_medium.sigma_a[2]);
        m.sigma_s = V3(fs->camera_medium.sigma_s[0], fs->camera_medium.sigma_s[1], fs->camera_medium.sigma_s[2]);
        m.sigma_t = m.sigma_a + m.sigma_s;
        m.g = fs->camera_medium.g;
        sc.viewMedium = (int)sc.media.size();
        sc.media.push_back(m);
    }
    for (int i = 0; i < fs->n_primitives; i++) {
        const PtrPrimitive &pp = fs->primitives[i];
        Primitive pr;
        Shape &sh = pr.shape;
        memset(&sh, 0, sizeof(Shape));
        sh.type = pp.shape_type;
        sh.ObjectToWorld = XfFromFlat(pp.to_world.m, pp.to_world.mInv);
        sh.reverseOrientation = pp.reverse_orientation != 0;
        sh.geom = pp.geom;
        bool ok = true;
        switch (pp.shape_type) {
        case SPHERE: /* shape.h:74-78 */
            sh.radius = pp.params[0];
            sh.zMin = -sh.radius;
            sh.zMax = sh.radius;
            sh.thetaMin = acosf(-1.f);
            sh.thetaMax = acosf(1.f);
            sh.phiMax = Radians(360);
            break;
        case SPHERE_PROCEDURAL: sh.radius = pp.params[0]; sh.reverseOrientation = false; break;
        case RECTANGLE: sh.sizex = Max(0, pp.params[0]); sh.sizey = Max(0, pp.params[1]); break; /* shape.h:119-122 */
        case DISK: /* shape.h:96-99 */
            sh.height = pp.params[0];
            sh.radius = pp.params[1];
            sh.innerRadius = pp.params[2];
            sh.phiMax = Radians(Clampf(pp.params[3], 0, 360));
            break;
        case BOX_PROCEDURAL: sh.length = V3(pp.params[0], pp.params[1], pp.params[2]); sh.reverseOrientation = false; break;
        case COMPONENT_PROCEDURAL:
            sh.toyBounds.pMin = V3(pp.params[0], pp.params[1], pp.params[2]);
            sh.toyBounds.pMax = V3(pp.params[3], pp.params[4], pp.params[5]);
            sh.toyId = pp.geom;
            sh.reverseOrientation = false;
            break;
        case MESH: { /* Mesh ctor: Shape(toWorld) with the mesh's own transform */
            const PtrMesh &pm = fs->meshes[pp.geom];
            sh.ObjectToWorld = XfFromFlat(pm.to_world.m, pm.to_world.mInv);
            sh.reverseOrientation = false;
        } break;
        case PARTICLECLOUD: /* Shape(Transform()) */
            sh.ObjectToWorld = XfTranslate(V3(0, 0, 0));
            sh.radius = sc.clouds[pp.geom].radius;
            sh.reverseOrientation = false;
            break;
        default: ok = false; break; /* "Skipping unknown shape type", scene.cpp:911-915 */
        }
        if (!ok) continue;
        sh.WorldToObject = XfInverse(sh.ObjectToWorld);
        sh.transformSwapsHandedness = XfSwapsHandedness(sh.ObjectToWorld);

        pr.material = -1;
        pr.emissive = false;
        pr.light = -1;
        int medium = -1;
        if (pp.mat.is_emissive && !pp.no_mat) {
            pr.emissive = true;
            pr.L = V3(pp.mat.textures[0].spec[0], pp.mat.textures[0].spec[1], pp.mat.textures[0].spec[2]);
        } else {
            if (!pp.no_mat && pp.mat.type >= Matte && pp.mat.type <= Disney) {
                Material m;
                m.type = pp.mat.type;
                m.ntex = pp.mat.n_textures;
                m.id = pp.mat.id;
                for (int k = 0; k < 16; k++) { m.tex[k] = conv_tex(pp.mat.textures[k]); m.vals[k] = pp.mat.vals[k]; }
                m.bssrdfTable = -1;
                if (m.type == Subsurface || m.type == KdSubsurface) {
                    BSSRDFTable t;
                    t.nRhoSamples = 100;
                    t.nRadiusSamples = 64;
                    if (m.type == Subsurface) ComputeBeamDiffusionBSSRDF(m.vals[2], m.vals[1], &t); /* subsurface.cpp:15-16 */
                    else ComputeBeamDiffusionBSSRDF(0, m.vals[0], &t);                              /* kdsubsurface.cpp:33 */
                    m.bssrdfTable = (int)sc.tables.size();
                    sc.tables.push_back(t);
                }
                pr.material = (int)sc.materials.size();
                sc.materials.push_back(m);
            }
            if (pp.medium.is_valid) {
                Medium m;
                m.sigma_a = V3(pp.medium.sigma_a[0], pp.medium.sigma_a[1], pp.medium.sigma_a[2]);
                m.sigma_s = V3(pp.medium.sigma_s[0], pp.medium.sigma_s[1], pp.medium.sigma_s[2]);
                m.sigma_t = m.sigma_a + m.sigma_s;
                m.g = pp.medium.g;
                medium = (int)sc.media.size();
                sc.media.push_back(m);
            }
        }
        pr.mediumInterface = MediumInterface(medium, -1);
        sc.prims.push_back(pr);
    }

    /* Aggregator::Wrap, primitive.cpp:620-642: mesh BVHs, cloud BVHs, then the top level */
    for (size_t i = 0; i < sc.prims.size(); i++) {
        if (sc.prims[i].shape.type != MESH) continue;
        MeshData &m = sc.meshes[sc.prims[i].shape.geom];
        if (!m.bvh.nodes.empty()) continue;
        std::vector<PrimitiveHandle> h(m.nTriangles);
        for (int t = 0; t < m.nTriangles; t++) { /* MeshGetBounds, mesh.cpp:609-626 */
            V3 p0 = m.p[m.indices[9 * t + 0]], p1 = m.p[m.indices[9 * t + 3]], p2 = m.p[m.indices[9 * t + 6]];
            Bounds3 b(p0, p1);
            b = Union(b, p2);
            h[t].bound = b;
            h[t].handle = t;
        }
        BuildBVH(m.bvh, h, 20, rng);
    }
    for (size_t i = 0; i < sc.prims.size(); i++) {
        if (sc.prims[i].shape.type != PARTICLECLOUD) continue;
        CloudData &c = sc.clouds[sc.prims[i].shape.geom];
        if (!c.bvh.nodes.empty()) continue;
        std::vector<PrimitiveHandle> h(c.nParticles);
        for (int t = 0; t < c.nParticles; t++) { /* ParticleCloudGetBounds, particlecloud.cpp:284-293 */
            Xf toWorld = XfTranslate(c.positions[t]);
            Bounds3 b(V3(-c.radius), V3(c.radius));
            h[t].bound = xf_bounds(toWorld, b);
            h[t].handle = t;
        }
        BuildBVH(c.bvh, h, 20, rng);
    }
    {
        std::vector<PrimitiveHandle> h(sc.prims.size());
        for (size_t i = 0; i < sc.prims.size(); i++) { /* MakeSceneTable, primitive.cpp:584-592 */
            h[i].bound = ShapeGetBounds(sc, sc.prims[i].shape);
            h[i].handle = (int)i;
        }
        if (!h.empty()) BuildBVH(sc.top, h, 20, rng);
    }

    /* lights: Aggregator::Insert (primitive.cpp:164-178) then MakeLights / SetLights (:205-244) */
    for (size_t i = 0; i < sc.prims.size(); i++) {
        if (!sc.prims[i].emissive) continue;
        Light l;
        l.type = DiffuseArea;
        l.flags = LF_Area;
        l.LightToWorld = sc.prims[i].shape.ObjectToWorld;
        l.WorldToLight = XfInverse(l.LightToWorld);
        l.Lemit = sc.prims[i].L;
        l.prim = (int)i;
        l.twoSided = true; /* Init_DiffuseArea(pri->Le(), pri->shape, true) */
        l.worldRadius = 0;
        sc.prims[i].light = (int)sc.lights.size();
        sc.lights.push_back(l);
    }
    sc.has_env = fs->has_envmap != 0;
    if (sc.has_env) {
        const PtrEnvMap &e = fs->envmap;
        EnvMap &env = sc.env;
        env.width = e.width;
        env.height = e.height;
        env.nu = e.nu;
        env.nv = e.nv;
        env.texels.resize((size_t)e.width * e.height);
        for (size_t k = 0; k < env.texels.size(); k++) env.texels[k] = V3(e.texels[3 * k], e.texels[3 * k + 1], e.texels[3 * k + 2]);
        env.func.assign(e.func, e.func + (size_t)e.nu * e.nv);
        env.cdf.assign(e.cdf, e.cdf + (size_t)(e.nu + 1) * e.nv);
        env.funcInt.assign(e.func_int, e.func_int + e.nv);
        env.margCdf.assign(e.marg_cdf, e.marg_cdf + e.nv + 1);
        env.margInt = e.marg_int;
        Light l;
        l.type = Infinite;
        l.flags = LF_Infinite;
        l.LightToWorld = XfFromFlat(e.light_to_world.m, e.light_to_world.mInv);
        l.WorldToLight = XfInverse(l.LightToWorld);
        l.prim = -1;
        l.twoSided = false;
        BoundingSphere(sc.top.nodes[0].bound, &l.worldCenter, &l.worldRadius); /* infinite.cpp:52-56 */
        sc.lights.push_back(l);
    }
    /* Aggregator::InsertDistantLight (primitive.cpp:180-189) -> Init_Distant (light.cpp:58-63), Distant_Prepare (distant.cpp:9-13).
     * The reference's scene API never calls it; the flat scene lists such lights behind the others (include/ptr_api.h). */
    for (int i = 0; i < fs->n_distant; i++) {
        const PtrDistantLight &d = fs->distant[i];
        Light l;
        l.type = Distant;
        l.flags = LF_DeltaDirection;
        l.LightToWorld = XfIdentity(); /* desc.toWorld = Transform() */
        l.WorldToLight = XfInverse(l.LightToWorld);
        l.Lemit = V3(d.L[0], d.L[1], d.L[2]);
        l.prim = -1;
        l.twoSided = false;
        l.wLight = Normalize(xf_vec(l.LightToWorld, V3(d.w[0], d.w[1], d.w[2])));
        BoundingSphere(sc.top.nodes[0].bound, &l.worldCenter, &l.worldRadius);
        sc.lights.push_back(l);
    }
    return os;
}

void orc_scene_destroy(void *h) { delete (OrcScene *)h; }

/* Aggregator::Intersect over a ray batch (SoA planes as in ptr_trace_closest).  order_dep[i]
 * (optional) = 1 when a brute-force closest hit over all primitives (each tested against the
 * full ray extent) names a different primitive/face or t than the reference traversal: the
 * rays on which the reference's answer depends on its own BVH order (ties). */
int orc_trace_closest(void *h, int n, const float *o, const float *d, const float *tmax, int32_t *prim, int32_t *face, float *t,
                      float *b0, float *b1, int32_t *order_dep) {
    OrcScene *os = (OrcScene *)h;
    const Scene &sc = os->sc;
    memset(&g_ctr, 0, sizeof(Counters));
    for (int i = 0; i < n; i++) {
        Ray r(V3(o[i], o[n + i], o[2 * n + i]), V3(d[i], d[n + i], d[2 * n + i]), tmax[i]);
        SurfaceInteraction si;
        Float bary[2] = {0, 0};
        bool hit = SceneIntersect(sc, r, &si, bary);
        prim[i] = hit ? si.primitive : -1;
        face[i] = hit ? si.faceIndex : -1;
        t[i] = hit ? r.tMax : tmax[i];
        if (b0) b0[i] = hit ? bary[0] : 0;
        if (b1) b1[i] = hit ? bary[1] : 0;
        if (order_dep) {
            int bp = -1, bf = -1;
            Float bt = tmax[i];
            int nties = 0;
            for (size_t p = 0; p < sc.prims.size(); p++) {
                Ray r2(r.o, r.d, tmax[i]);
                Float th;
                SurfaceInteraction s2;
                if (ShapeIntersect(sc, (int)p, r2, &th, &s2, nullptr)) {
                    if (bp < 0 || th < bt) { bt = th; bp = (int)p; bf = s2.faceIndex; nties = 0; }
                    else if (th == bt) nties++;
                }
            }
            order_dep[i] = (bp != prim[i]) || (hit && (bt != t[i] || bf != face[i])) || nties > 0;
        }
    }
    add_counters(os->total, g_ctr);
    return 0;
}

int orc_trace_occluded(void *h, int n, const float *o, const float *d, const float *tmax, int32_t *occluded) {
    OrcScene *os = (OrcScene *)h;
    for (int i = 0; i < n; i++) {
        Ray r(V3(o[i], o[n + i], o[2 * n + i]), V3(d[i], d[n + i], d[2 * n + i]), tmax[i]);
        SurfaceInteraction si;
        occluded[i] = SceneIntersect(os->sc, r, &si) ? 1 : 0;
    }
    return 0;
}

static Camera conv_cam(const PtrCamera *c) {
    Camera k;
    k.position = V3(c->position[0], c->position[1], c->position[2]);
    k.lower_left = V3(c->lower_left[0], c->lower_left[1], c->lower_left[2]);
    k.horizontal = V3(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
    k.vertical = V3(c->vertical[0], c->vertical[1], c->vertical[2]);
    k.u = V3(c->u[0], c->u[1], c->u[2]);
    k.v = V3(c->v[0], c->v[1], c->v[2]);
    k.w = V3(c->w[0], c->w[1], c->w[2]);
    k.lensRadius = c->lens_radius;
    return k;
}

/* camera half of Render, main.cpp:436-443: draws 0..3 of stream (pixel, sample) */
static Ray gen_ray(const Scene &sc, const Camera &cam, int w, int hgt, int i, int j, RNG &state, Float *weight) {
    Float u0 = (Float)i / (Float)w;
    Float u1 = (Float)j / (Float)hgt;
    Float u = ((Float)i + state.next()) / (Float)w;
    Float v = ((Float)j + state.next()) / (Float)hgt;
    Float py = state.next(), px = state.next(); /* rand_point2: Point2f(rand, rand), right-to-left */
    V2 sample = ConcentricSampleDisk(V2(px, py));
    Ray ray = CameraSpawnRay(cam, u, v, sample, sc.viewMedium);
    *weight = GetFilterWeight(u0, u1, u, v);
    return ray;
}

int orc_generate_rays(void *h, const PtrCamera *pc, int w, int hgt, uint64_t seed, int sample, float *o, float *d, float *weight) {
    OrcScene *os = (OrcScene *)h;
    Camera cam = conv_cam(pc);
    int n = w * hgt;
    for (int j = 0; j < hgt; j++)
        for (int i = 0; i < w; i++) {
            int p = j * w + i;
            RNG st = MakeRNG(seed, (uint32_t)p, (uint32_t)sample);
            Float wt;
            Ray r = gen_ray(os->sc, cam, w, hgt, i, j, st, &wt);
            o[p] = r.o.x; o[n + p] = r.o.y; o[2 * n + p] = r.o.z;
            d[p] = r.d.x; d[n + p] = r.d.y; d[2 * n + p] = r.d.z;
            weight[p] = wt;
        }
    return 0;
}

/* Render<<<>>> restated (main.cpp:421-457) for samples [sample_begin, sample_begin+ns) of every
 * pixel, accumulating into we/acc_weight/samples exactly as Pixel does.  The camera draws come
 * from a COPY of the stream state while the integrator restarts at draw 0 (main.cpp:433 quirk). */
int orc_render(void *h, const PtrCamera *pc, int w, int hgt, uint64_t seed, int sample_begin, int ns, int integrator, float *we,
               double *acc_weight, int64_t *samples, int nthreads) {
    OrcScene *os = (OrcScene *)h;
    const Scene &sc = os->sc;
    Camera cam = conv_cam(pc);
    if (nthreads < 1) nthreads = 1;
    std::atomic<int> next_row(0);
    std::vector<Counters> ctrs(nthreads);
    auto work = [&](int tid) {
        memset(&g_ctr, 0, sizeof(Counters));
        for (;;) {
            int j = next_row.fetch_add(1);
            if (j >= hgt) break;
            for (int i = 0; i < w; i++) {
                int p = j * w + i;
                V3 out(we[3 * p], we[3 * p + 1], we[3 * p + 2]);
                for (int s = sample_begin; s < sample_begin + ns; s++) {
                    RNG cam_state = MakeRNG(seed, (uint32_t)p, (uint32_t)s);
                    RNG pix_state = cam_state;
                    Float wt;
                    Ray ray = gen_ray(sc, cam, w, hgt, i, j, cam_state, &wt);
                    g_ctr.camera_rays++;
                    V3 L = Li(sc, integrator, ray, pix_state);
                    out += wt * L;
                    acc_weight[p] += wt;
                    samples[p]++;
                }
                we[3 * p] = out.x; we[3 * p + 1] = out.y; we[3 * p + 2] = out.z;
            }
        }
        ctrs[tid] = g_ctr;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    for (int t = 0; t < nthreads; t++) add_counters(os->total, ctrs[t]);
    return 0;
}

void orc_counters_read(void *h, PtrCounters *out) {
    OrcScene *os = (OrcScene *)h;
    static_assert(sizeof(PtrCounters) == sizeof(Counters), "counter layouts must match");
    memcpy(out, &os->total, sizeof(Counters));
}
void orc_counters_reset(void *h) { memset(&((OrcScene *)h)->total, 0, sizeof(Counters)); }

/* reference-BVH statistics: nodes and bytes as the reference would allocate them (Node 56 B,
 * PrimitiveHandle 28 B) */
void orc_bvh_info(void *h, int64_t *n_nodes, int64_t *n_handles) {
    OrcScene *os = (OrcScene *)h;
    int64_t nn = os->sc.top.nodes.size(), nh = os->sc.top.handles.size();
    for (auto &m : os->sc.meshes) { nn += m.bvh.nodes.size(); nh += m.bvh.handles.size(); }
    for (auto &c : os->sc.clouds) { nn += c.bvh.nodes.size(); nh += c.bvh.handles.size(); }
    *n_nodes = nn;
    *n_handles = nh;
}

/* glibc rand() restatement check hook */
int orc_glibc_rand(int n_skip) {
    GlibcRand r(1);
    for (int i = 0; i < n_skip; i++) r.next();
    return r.next();
}
float orc_rng_draw(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
    RNG r = MakeRNG(seed, pixel, sample);
    r.dim = dim;
    return r.next();
}

/* ---- unit hooks: BSDF f / Pdf / Sample_f of primitive `prim`'s material at a canonical
 * interaction (n = +z, dpdu = +x, uv given), directions in WORLD space ------------------- */
static SurfaceInteraction unit_si(int prim, float u, float v) {
    SurfaceInteraction si;
    si.p = V3(0, 0, 0);
    si.n = V3(0, 0, 1);
    si.dpdu = V3(1, 0, 0);
    si.dpdv = V3(0, 1, 0);
    si.uv = V2(u, v);
    si.primitive = prim;
    si.shape = prim;
    return si;
}
int orc_bsdf_eval(void *h, int prim, int n, const float *wo, const float *wi, const float *uv, float *f, float *pdf) {
    OrcScene *os = (OrcScene *)h;
    for (int i = 0; i < n; i++) {
        SurfaceInteraction si = unit_si(prim, uv[2 * i], uv[2 * i + 1]);
        BSDF bsdf(si);
        ComputeScatteringFunctions(os->sc, &si, &bsdf);
        V3 o(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), w(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
        V3 r = bsdf.f(o, w, BSDF_ALL & ~BSDF_SPECULAR);
        f[3 * i] = r.x; f[3 * i + 1] = r.y; f[3 * i + 2] = r.z;
        pdf[i] = bsdf.Pdf(o, w, BSDF_ALL & ~BSDF_SPECULAR);
    }
    return 0;
}
int orc_bsdf_sample(void *h, int prim, int n, const float *wo, const float *u2, const float *uv, float *wi, float *f, float *pdf,
                    int32_t *flags) {
    OrcScene *os = (OrcScene *)h;
    for (int i = 0; i < n; i++) {
        SurfaceInteraction si = unit_si(prim, uv[2 * i], uv[2 * i + 1]);
        BSDF bsdf(si);
        ComputeScatteringFunctions(os->sc, &si, &bsdf);
        V3 o(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), w(0, 0, 0);
        Float p = 0;
        int fl = 0;
        V3 r = bsdf.Sample_f(o, &w, V2(u2[2 * i], u2[2 * i + 1]), &p, BSDF_ALL, &fl);
        wi[3 * i] = w.x; wi[3 * i + 1] = w.y; wi[3 * i + 2] = w.z;
        f[3 * i] = r.x; f[3 * i + 1] = r.y; f[3 * i + 2] = r.z;
        pdf[i] = p;
        flags[i] = fl;
    }
    return 0;
}

/* ---- unit hooks: scalar building blocks, n records of `in`, results to `out` (widths per kind below) ----
 *  kind 0 fr_dielectric   in: cosI etaI etaT                      out: F
 *  kind 1 fr_conductor    in: cosI etaI(3) etaT(3) k(3)           out: F(3)
 *  kind 2 ggx_d           in: ax ay wh(3)                         out: D
 *  kind 3 ggx_g1          in: ax ay w(3)                          out: G1
 *  kind 4 ggx_pdf         in: ax ay wo(3) wh(3)                   out: pdf
 *  kind 5 ggx_sample_wh   in: ax ay wo(3) u(2)                    out: wh(3)
 *  kind 6 hg_p            in: cosTheta g                          out: p
 *  kind 7 hg_sample       in: g wo(3) u(2)                        out: wi(3) p
 *  kind 8 rough_to_alpha  in: roughness                           out: alpha
 *  kind 9 env_sample      in: u(2)               (scene h)        out: uv(2) pdf
 *  kind 10 env_pdf        in: uv(2)              (scene h)        out: pdf
 *  kind 11 sss_sr         in: r                  (scene h, prim)  out: Sr(3)
 *  kind 12 sss_pdf_sr     in: ch r               (scene h, prim)  out: pdf
 *  kind 13 sss_sample_sr  in: ch u               (scene h, prim)  out: r
 * Returns 0, or -1 for an unknown kind / missing scene part. */
int orc_unit_eval(void *h, int prim, int kind, int n, const float *in, int in_w, float *out, int out_w) {
    OrcScene *os = (OrcScene *)h;
    BSDF sss_bsdf(unit_si(prim < 0 ? 0 : prim, 0.5f, 0.5f));
    if (kind >= 9 && !os) return -1;
    if (kind == 9 || kind == 10) { if (os->sc.env.func.empty()) return -1; }
    if (kind >= 11) {
        SurfaceInteraction si = unit_si(prim, 0.5f, 0.5f);
        ComputeScatteringFunctions(os->sc, &si, &sss_bsdf);
        if (!sss_bsdf.has_bssrdf) return -1;
    }
    for (int i = 0; i < n; i++) {
        const float *a = in + (size_t)i * in_w;
        float *o = out + (size_t)i * out_w;
        switch (kind) {
        case 0: o[0] = Unit_FrDielectric(a[0], a[1], a[2]); break;
        case 1: { V3 r = Unit_FrConductor(a[0], V3(a[1], a[2], a[3]), V3(a[4], a[5], a[6]), V3(a[7], a[8], a[9])); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case 2: o[0] = Unit_GGX_D(a[0], a[1], V3(a[2], a[3], a[4])); break;
        case 3: o[0] = Unit_GGX_G1(a[0], a[1], V3(a[2], a[3], a[4])); break;
        case 4: o[0] = Unit_GGX_Pdf(a[0], a[1], V3(a[2], a[3], a[4]), V3(a[5], a[6], a[7])); break;
        case 5: { V3 r = Unit_GGX_Sample_wh(a[0], a[1], V3(a[2], a[3], a[4]), V2(a[5], a[6])); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case 6: o[0] = Unit_HG(a[0], a[1]); break;
        case 7: { PhaseFunction ph; ph.g = a[0]; V3 wi; Float p = PhaseSample_p(ph, V3(a[1], a[2], a[3]), &wi, V2(a[4], a[5])); o[0] = wi.x; o[1] = wi.y; o[2] = wi.z; o[3] = p; } break;
        case 8: o[0] = RoughnessToAlpha(a[0]); break;
        case 9: { Float pdf = 0; V2 uv = Unit_D2_Sample(os->sc.env, V2(a[0], a[1]), &pdf); o[0] = uv[0]; o[1] = uv[1]; o[2] = pdf; } break;
        case 10: o[0] = Unit_D2_Pdf(os->sc.env, V2(a[0], a[1])); break;
        case 11: { V3 r = Unit_Sr(sss_bsdf.bssrdf, a[0]); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
        case 12: o[0] = Unit_Pdf_Sr(sss_bsdf.bssrdf, (int)a[0], a[1]); break;
        case 13: o[0] = Unit_Sample_Sr(sss_bsdf.bssrdf, (int)a[0], a[1]); break;
        default: return -1;
        }
    }
    return 0;
}

} /* extern "C" */
