"""the geometry BVHs built on the device (rt_build.hip: Morton codes, radix sort, Karras hierarchy, refit,
collapse) against the host binned-SAH build of the same scene.  The traversal and the leaf tests are the same code for both
trees, so the nearest hit distance of every ray must be the same float, and the ids the same wherever the nearest hit is
unique; the oracle (which follows the reference's own builders) is the third voice on a sample."""
import numpy as np
import pytest

import pathtracer_amd as pa

import common

pytestmark = pytest.mark.gpu
FMAX = np.finfo(np.float32).max

CASES = {
    "knot": ("knot", 96, 0, 256, 144),
    "fluid": ("fluid", 60000, 0, 256, 144),
    "room": ("room", 24, 0, 256, 144),
    "subsurface": ("subsurface", 64, 0, 256, 144),
    "zoo": ("zoo", 0, 0, 256, 144),           # meshes of a handful of triangles next to analytic shapes: single-leaf trees
    "knot_full": ("knot", 660, 0, 640, 360),         # configs[1]: 871,200 triangles
    "fluid_full": ("fluid", 4000000, 0, 640, 360),   # configs[3]: 4 M particles, the geometry that changes per frame
}


@pytest.fixture(scope="module", params=list(CASES))
def pair(cuda, request):
    fam, p0, p1, W, H = CASES[request.param]
    sc = pa.BuiltScene(fam, W, H, p0, p1)
    host, dev = pa.Renderer(0), pa.Renderer(0)
    try:
        pa.set_bvh_build_mode("host")
        host.upload(sc)
        pa.set_bvh_build_mode("device")
        dev.upload(sc)
    finally:
        pa.set_bvh_build_mode("host")
    for r in (host, dev):
        r.set_camera(sc.camera)
        r.film_init(W, H, 3)
    yield sc, host, dev, W, H
    host.close()
    dev.close()
    sc.close()


def rays_of(sc, r, W, H):
    o, d, _ = common.gpu_generate_rays(r, W * H, 1)
    lo = 4.0 if sc.name != "fluid" else 2.0
    o2, d2 = common.random_rays(60000, 11, -lo, lo)
    return np.concatenate([o, o2], 1), np.concatenate([d, d2], 1)


def test_same_items_different_tree(pair):
    sc, host, dev, W, H = pair
    a, b = host.bvh_info(), dev.bvh_info()
    assert a["items"] == b["items"]
    assert host.bvh_build_info()["builder"] == "host_sah" and dev.bvh_build_info()["builder"] == "device_lbvh"
    assert dev.bvh_build_info()["depth"] <= 23  # 3 stack entries per level + 24 <= BVH_STACK
    if a["items"] > 64:
        hi, di = host.bvh_build_info(), dev.bvh_build_info()
        assert 0 < b["nodes"] < b["items"] and di["build_ms"] > 0
        # the surface-area estimate of the traversal work: the SAH tree is the better one (a binned-SAH build that falls
        # back to median splits loses this comparison by 40 %), the linear tree within 1.75x of it (measured: 1.17-1.2x on
        # the meshes, 1.45-1.5x on clouds of overlapping particle spheres, which spatial-median splits cut through)
        assert hi["sah_boxes"] + hi["sah_items"] <= 1.02 * (di["sah_boxes"] + di["sah_items"])
        assert di["sah_boxes"] + di["sah_items"] <= 1.75 * (hi["sah_boxes"] + hi["sah_items"])
    if a["items"] > 500000:
        assert dev.bvh_build_info()["build_ms"] < 0.1 * host.bvh_build_info()["build_ms"]


def test_closest_hits_match_host_tree(pair):
    sc, host, dev, W, H = pair
    o, d = rays_of(sc, host, W, H)
    tmax = np.full(o.shape[1], FMAX, np.float32)
    a = common.gpu_trace_closest(host, o, d, tmax)
    b = common.gpu_trace_closest(dev, o, d, tmax)
    assert (a["prim"] >= 0).mean() > 0.1
    assert np.array_equal(a["prim"] >= 0, b["prim"] >= 0)
    assert np.array_equal(a["t"], b["t"])                       # the nearest distance does not depend on the tree
    same = (a["prim"] == b["prim"]) & (a["face"] == b["face"])  # ids differ only where two items are hit at the same t
    assert same.mean() > 0.999
    assert np.array_equal(a["b0"][same], b["b0"][same]) and np.array_equal(a["b1"][same], b["b1"][same])


def test_occlusion_matches_host_tree(pair):
    sc, host, dev, W, H = pair
    o, d = rays_of(sc, host, W, H)
    rng = np.random.default_rng(5)
    tmax = rng.uniform(0.5, 12.0, o.shape[1]).astype(np.float32)
    assert np.array_equal(common.gpu_trace_occluded(host, o, d, tmax), common.gpu_trace_occluded(dev, o, d, tmax))


def test_against_the_oracle(pair):
    sc, host, dev, W, H = pair
    o, d = rays_of(sc, host, W, H)
    idx = np.arange(0, o.shape[1], 23)
    o, d = np.ascontiguousarray(o[:, idx]), np.ascontiguousarray(d[:, idx])
    tmax = np.full(idx.size, FMAX, np.float32)
    oracle = common.oracle_of(sc)
    c = oracle.trace_closest(o, d, tmax)
    oracle.close()
    g = common.gpu_trace_closest(dev, o, d, tmax)
    ok = c["order_dep"] == 0
    assert ok.mean() > 0.98
    assert np.array_equal(g["prim"][ok], c["prim"][ok]) and np.array_equal(g["face"][ok], c["face"][ok])
    hit = ok & (c["prim"] >= 0)
    assert np.max(np.abs(g["t"][hit] - c["t"][hit]) / np.abs(c["t"][hit])) <= 1e-5


def test_render_is_the_same_image(pair):
    """The film of a render: the estimator only sees hits, so the two trees give the same film except where equal-t
    ties picked another face (a handful of pixels)."""
    sc, host, dev, W, H = pair
    films = []
    for r in (host, dev):
        r.film_init(W, H, 3)
        r.render(2, pa.LI_VOLPATH)
        films.append(r.film()[0])
    differ = (films[0] != films[1]).any(axis=-1).mean() if films[0].ndim == 3 else (films[0] != films[1]).mean()
    assert differ < 5e-3
