"""Edge cases of the hot path, HIP path against the oracle through the C-ABI, with the geometry trees built on the host
(SAH) and on the device (LBVH): empty scene, meshes of one to five triangles, zero-area and duplicated triangles, a
thousand coincident triangles (all Morton codes equal), clouds of one particle / coincident particles / zero radius,
axis-parallel rays, rays born on surfaces, zero and tiny extents."""
import numpy as np
import pytest

import pathtracer_amd as pa
from oracle import orc
from pathtracer_amd import scene as S

import common

pytestmark = pytest.mark.gpu
FMAX = np.finfo(np.float32).max
MATTE = dict(kd=(0.6, 0.5, 0.4))


def compare(sc, rays_o, rays_d, tmax, builder, same_face=None):
    o = orc.OracleScene(sc.flat, 0)
    pa.set_bvh_build_mode(builder)
    try:
        r = pa.Renderer(0)
        r.upload(sc)
    finally:
        pa.set_bvh_build_mode("host")
    c = o.trace_closest(rays_o, rays_d, tmax)
    g = common.gpu_trace_closest(r, rays_o, rays_d, tmax)
    ok = c["order_dep"] == 0
    gf, cf = g["face"].copy(), c["face"].copy()
    if same_face == "all":  # every item is the same item: any of them may be reported
        gf[:] = 0
        cf[:] = 0
        same_face = None
    for a, b in (same_face or {}).items():  # faces that are the same triangle twice: either may be reported
        gf[gf == a] = b
        cf[cf == a] = b
    assert np.array_equal(g["prim"][ok], c["prim"][ok]) and np.array_equal(gf[ok], cf[ok])
    assert np.array_equal(g["prim"] >= 0, c["prim"] >= 0)
    hit = c["prim"] >= 0
    assert np.array_equal(g["t"][hit], c["t"][hit])  # the same float, ties included: the nearest distance does not depend on order
    occ_c = o.trace_occluded(rays_o, rays_d, tmax)
    occ_g = common.gpu_trace_occluded(r, rays_o, rays_d, tmax)
    assert np.array_equal(occ_c != 0, occ_g != 0)
    r.close()
    o.close()
    return c


def grid_rays(n=48, z=5.0, half=1.5):
    """parallel rays down -z through a square, plus the same set tilted"""
    u = np.linspace(-half, half, n, dtype=np.float32)
    x, y = np.meshgrid(u, u)
    o = np.stack([x.ravel(), y.ravel(), np.full(n * n, z, np.float32)]).astype(np.float32)
    d = np.stack([np.zeros(n * n), np.zeros(n * n), -np.ones(n * n)]).astype(np.float32)  # axis-parallel: two zero components
    o2, d2 = common.random_rays(n * n, 3, -2.0, 2.0)
    return np.concatenate([o, o2], 1), np.concatenate([d, d2], 1)


@pytest.mark.parametrize("builder", ["host", "device"])
@pytest.mark.parametrize("ntri", [1, 2, 3, 4, 5, 9])
def test_tiny_meshes(cuda, builder, ntri):
    rng = np.random.default_rng(ntri)
    v = rng.uniform(-1, 1, (3 * ntri, 3)).astype(np.float32)
    t = np.arange(3 * ntri, dtype=np.int32).reshape(ntri, 3)
    b = pa.FlatSceneBuilder()
    b.mesh(v, t, pa.material("matte", **MATTE))
    sc = b.finish()
    o, d = grid_rays()
    c = compare(sc, o, d, np.full(o.shape[1], FMAX, np.float32), builder)
    assert (c["prim"] >= 0).any()


@pytest.mark.parametrize("builder", ["host", "device"])
def test_degenerate_and_duplicated_triangles(cuda, builder):
    v = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0],        # a proper triangle ...
                  [-1, -1, 0], [1, -1, 0], [0, 1, 0],        # ... twice: every hit on it is a tie
                  [0.5, 0.5, 0.5], [0.5, 0.5, 0.5], [0.5, 0.5, 0.5],   # a point
                  [-1, 0, 1], [0, 0, 1], [1, 0, 1],          # a segment (collinear corners)
                  [-1, -1, -1], [1, -1, -1], [0, 1, -1]], np.float32)
    t = np.arange(15, dtype=np.int32).reshape(5, 3)
    b = pa.FlatSceneBuilder()
    b.mesh(v, t, pa.material("matte", **MATTE))
    sc = b.finish()
    o, d = grid_rays()
    # (the oracle flags a tie only when its own walk meets both copies; a copy whose leaf box is culled at t == tMax is not met)
    c = compare(sc, o, d, np.full(o.shape[1], FMAX, np.float32), builder, same_face={1: 0})
    assert (c["prim"] >= 0).any()


@pytest.mark.parametrize("builder", ["host", "device"])
def test_a_thousand_coincident_triangles(cuda, builder):
    """all centroids equal: the SAH builder falls back to median splits, the LBVH's keys differ only in the item index"""
    tri = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32)
    v = np.tile(tri, (1000, 1))
    t = np.arange(3000, dtype=np.int32).reshape(1000, 3)
    b = pa.FlatSceneBuilder()
    b.mesh(v, t, pa.material("matte", **MATTE))
    sc = b.finish()
    o, d = grid_rays(24)
    o_sc = orc.OracleScene(sc.flat, 0)
    pa.set_bvh_build_mode(builder)
    try:
        r = pa.Renderer(0)
        r.upload(sc)
    finally:
        pa.set_bvh_build_mode("host")
    tm = np.full(o.shape[1], FMAX, np.float32)
    c, g = o_sc.trace_closest(o, d, tm), common.gpu_trace_closest(r, o, d, tm)
    assert np.array_equal(g["prim"] >= 0, c["prim"] >= 0)
    hit = c["prim"] >= 0
    assert hit.any() and np.array_equal(g["t"][hit], c["t"][hit])  # which of the thousand is reported is order-dependent, where is not
    assert r.bvh_build_info()["depth"] <= 23
    r.close()
    o_sc.close()


@pytest.mark.parametrize("builder", ["host", "device"])
@pytest.mark.parametrize("case", ["one", "coincident", "line", "zero_radius"])
def test_particle_clouds(cuda, builder, case):
    if case == "one":
        pos, radius = np.array([[0.2, -0.1, 0.3]], np.float32), 0.7
    elif case == "coincident":
        pos, radius = np.tile(np.array([[0.0, 0.0, 0.0]], np.float32), (37, 1)), 0.5
    elif case == "line":
        pos = np.stack([np.linspace(-1.5, 1.5, 64), np.zeros(64), np.zeros(64)], 1).astype(np.float32)
        radius = 0.08
    else:
        pos, radius = np.random.default_rng(2).uniform(-1, 1, (50, 3)).astype(np.float32), 0.0
    b = pa.FlatSceneBuilder()
    b.cloud(pos, radius, pa.material("matte", **MATTE))
    sc = b.finish()
    o, d = grid_rays()
    c = compare(sc, o, d, np.full(o.shape[1], FMAX, np.float32), builder, same_face="all" if case == "coincident" else None)
    if case != "zero_radius":
        assert (c["prim"] >= 0).any()


def test_empty_scene(cuda):
    b = pa.FlatSceneBuilder()
    sc = b.finish()
    o, d = grid_rays(16)
    tm = np.full(o.shape[1], FMAX, np.float32)
    with pa.Renderer(0) as r:
        r.upload(sc)
        g = common.gpu_trace_closest(r, o, d, tm)
        assert (g["prim"] == -1).all() and np.array_equal(g["t"], tm)
        assert (common.gpu_trace_occluded(r, o, d, tm) == 0).all()
        r.set_camera(pa.look_at_camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, 4 / 3))
        r.film_init(32, 24, 1)
        for integ in common.INTEGRATORS.values():
            r.render(2, integ)
        we, acc, smp = r.film()
        assert (we == 0).all() and (smp == 8).all() and (acc > 0).all()  # nothing to see, every sample still counted


def test_extents_and_origins(cuda):
    """zero and tiny extents, rays born on the surface they would hit, rays pointing away"""
    b = pa.FlatSceneBuilder()
    b.sphere(1.0, pa.material("matte", **MATTE))
    b.rectangle(4, 4, pa.material("matte", **MATTE), to_world=S.translate(0, 0, -2))
    v = np.array([[-1, -1, 2], [1, -1, 2], [0, 1, 2]], np.float32)
    b.mesh(v, np.array([[0, 1, 2]], np.int32), pa.material("matte", **MATTE))
    sc = b.finish()
    n = 4096
    rng = np.random.default_rng(4)
    dirs = rng.normal(size=(3, n)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=0, keepdims=True)
    on_sphere = dirs.copy()                                  # origins exactly on the unit sphere
    o = np.concatenate([on_sphere, np.zeros((3, n), np.float32)], 1)   # ... and at its centre
    d = np.concatenate([rng.normal(size=(3, n)), rng.normal(size=(3, n))], 1).astype(np.float32)
    d /= np.linalg.norm(d, axis=0, keepdims=True)
    for tmax in (FMAX, 1.0, 1e-3, 0.0):
        compare(sc, o, d, np.full(2 * n, tmax, np.float32), "host")


def test_non_finite_and_null_rays(cuda):
    """NaN / infinite / zero-length components in origins, directions and extents: every ray ends (the persistent tracer
    drains), and since the comparisons involved are IEEE comparisons on both sides the answers equal the oracle's"""
    sc = common.build("zoo", 64, 48)  # analytic shapes, SDF shapes, a mesh
    o_sc = common.oracle_of(sc)
    r = pa.Renderer(0)
    r.upload(sc)
    n = 4096
    o, d = common.random_rays(n, 9, -3.0, 3.0)
    tm = np.full(n, FMAX, np.float32)
    rng = np.random.default_rng(4)
    bad = [np.nan, np.inf, -np.inf, 0.0, -0.0, FMAX, -FMAX, 1e-45]
    for i in range(0, n, 2):  # every other ray gets one to three bad words
        for _ in range(int(rng.integers(1, 4))):
            which, c, v = int(rng.integers(0, 3)), int(rng.integers(0, 3)), np.float32(rng.choice(bad))
            if which == 0:
                o[c, i] = v
            elif which == 1:
                d[c, i] = v
            else:
                tm[i] = v
    d[:, 10] = 0  # a null direction
    c = o_sc.trace_closest(o, d, tm)
    g = common.gpu_trace_closest(r, o, d, tm)
    ok = c["order_dep"] == 0
    assert np.array_equal(g["prim"][ok], c["prim"][ok])
    hit = (c["prim"] >= 0) & ok
    assert np.array_equal(g["t"][hit], c["t"][hit])
    assert np.array_equal(o_sc.trace_occluded(o, d, tm) != 0, common.gpu_trace_occluded(r, o, d, tm) != 0)
    r.close()
    o_sc.close()
    sc.close()


@pytest.mark.parametrize("size", [(1, 1), (3, 2), (7, 5)])
def test_tiny_films_and_many_samples(cuda, size):
    """films of a few pixels with sample counts that straddle the pass size (64 samples of every pixel) and the six lanes:
    the sample counts are exact, a render split over calls is the single render bit for bit, the picture is the oracle's"""
    w, h = size
    sc = common.build("cornell", w, h)
    o_sc = common.oracle_of(sc)
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(sc.camera)
        for ns in (1, 63, 64, 65, 129, 777):
            r.film_init(w, h, 5)
            r.render(ns, pa.LI_PATHSAMPLED)
            we, acc, smp = r.film()
            assert np.all(smp == ns)
            one = we.copy()
            r.film_init(w, h, 5)
            left = ns
            for part in (1, 64, 2, 300, 10 ** 6):
                k = min(part, left)
                r.render(k, pa.LI_PATHSAMPLED)
                left -= k
                if left == 0:
                    break
            assert np.array_equal(r.film()[0], one)
        img = r.image()
        we, acc, osmp = o_sc.render(sc.camera, w, h, 5, 0, 777, pa.LI_PATHSAMPLED, 8)
        ref = orc.image_of(we, acc).reshape(h, w, 3)
        assert np.allclose(img, ref, rtol=2e-2, atol=1e-3), (img, ref)
        r.film_init(w, h, 5)
        r.render(0, pa.LI_PATH)  # nothing to do is not an error
        assert np.all(r.film()[2] == 0)
    o_sc.close()
    sc.close()
