"""The fast top-level tree (PTR_FAST_TOP=1; dev_trace.h "fast top level", ptr_build.cpp build_fast_top).

Rays whose answer provably does not need the visiting order of the reference's own tree over the primitives
(primitive.cpp:374-431, 537-582) walk a SAH tree over the same primitives; the closest-hit rays for which the walk cannot rule
out an order dependence are traced again through the reference's tree.  The claim is "the same answer for every ray": hits
(primitive, item, t as the same float), occlusion and films must be IDENTICAL with the switch on and off - and with it on the
oracle comparison of the other tests holds as well.  Off by default: exact but measured slower."""
import os

import numpy as np
import pytest

import pathtracer_amd as pa

import common

pytestmark = pytest.mark.gpu
W, H = 64, 48


def renderer(fast):
    old = os.environ.get("PTR_FAST_TOP")
    os.environ["PTR_FAST_TOP"] = "1" if fast else "0"
    try:
        return pa.Renderer(0)  # the switch is read when the context is created
    finally:
        if old is None:
            os.environ.pop("PTR_FAST_TOP", None)
        else:
            os.environ["PTR_FAST_TOP"] = old


def rays_for(sc, n, seed):
    """camera rays of the scene plus random unit rays through its bounds, a third of them bounded"""
    rng = np.random.default_rng(seed)
    o, d = common.random_rays(n, seed, -12.0, 12.0)
    tmax = np.full(n, np.float32(3.4028235e38), dtype=np.float32)
    tmax[::3] = rng.uniform(0.5, 20.0, size=tmax[::3].shape).astype(np.float32)
    return o, d, tmax


@pytest.mark.parametrize("name", ["room", "cornell", "cornell_glass", "zoo", "origami", "fluid", "volcaustic"])
def test_hits_identical(name):
    sc = common.build(name, W, H)
    out = []
    for fast in (False, True):
        r = renderer(fast)
        r.upload(sc)
        r.set_camera(sc.camera)
        r.film_init(W, H, 5)
        co, cd, _ = common.gpu_generate_rays(r, W * H, 0)
        o, d, tmax = rays_for(sc, 60000, 17)
        o = np.concatenate([co, o], axis=1)
        d = np.concatenate([cd, d], axis=1)
        tmax = np.concatenate([np.full(W * H, np.float32(3.4028235e38), dtype=np.float32), tmax])
        r.counters_reset()
        g = common.gpu_trace_closest(r, o, d, tmax)
        occ = common.gpu_trace_occluded(r, o, d, tmax)
        out.append((g, occ, r.pipeline_stats()["fast_top_deferred"]))
        r.close()
    (g0, occ0, def0), (g1, occ1, def1) = out
    assert def0 == 0
    print("%s: %d of %d closest-hit rays went back to the reference's tree" % (name, def1, g0["prim"].size))
    assert np.array_equal(g0["prim"], g1["prim"])
    assert np.array_equal(g0["face"], g1["face"])
    assert np.array_equal(g0["t"].view(np.uint32), g1["t"].view(np.uint32))
    assert np.array_equal(g0["b0"].view(np.uint32), g1["b0"].view(np.uint32))
    assert np.array_equal(occ0 != 0, occ1 != 0)
    assert def1 < 0.2 * g0["prim"].size  # the fallback is the exception
    # and the oracle agrees, ties aside
    oracle = common.oracle_of(sc)
    c = oracle.trace_closest(o, d, tmax)
    ok = c["order_dep"] == 0
    assert np.array_equal(g1["prim"][ok], c["prim"][ok])
    h = ok & (c["prim"] >= 0)
    assert np.array_equal(g1["t"][h], c["t"][h])
    sc.close()


def test_non_unit_rays_take_the_reference_tree():
    """an SDF shape measures t along the normalised direction (procedural.cpp:238): only unit directions may use the fast tree"""
    sc = common.build("room", W, H)
    o, d, tmax = rays_for(sc, 20000, 3)
    d = (d * np.float32(2.5)).astype(np.float32)
    res = []
    for fast in (False, True):
        r = renderer(fast)
        r.upload(sc)
        r.counters_reset()
        res.append((common.gpu_trace_closest(r, o, d, tmax), r.pipeline_stats()["fast_top_deferred"]))
        r.close()
    assert res[1][1] == o.shape[1]  # every ray deferred
    for k in ("prim", "face"):
        assert np.array_equal(res[0][0][k], res[1][0][k])
    assert np.array_equal(res[0][0]["t"].view(np.uint32), res[1][0]["t"].view(np.uint32))
    sc.close()


@pytest.mark.parametrize("name,integ", [("room", pa.LI_PATHSAMPLED), ("room", pa.LI_VOLPATH), ("cornell_fog", pa.LI_VOLPATH), ("zoo", pa.LI_PATH),
                                        ("origami", pa.LI_PATHSAMPLED), ("fluid", pa.LI_DIRECT)])
def test_film_identical(name, integ):
    sc = common.build(name, W, H)
    films = []
    for fast in (False, True):
        r = renderer(fast)
        r.upload(sc)
        r.set_camera(sc.camera)
        r.film_init(W, H, 21)
        r.render(6, integ)
        we, acc, smp = r.film()
        films.append(we.copy())
        r.close()
    assert np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32))
    sc.close()
