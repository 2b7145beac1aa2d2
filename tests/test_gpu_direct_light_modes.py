"""Li_VolPath's direct light where the only primitives without a material are emitters: VisibilityTester::Tr and
IntersectTr (light.cpp:10-40, primitive.cpp:433-461) cannot go past their first hit there, so the HIP path issues them as
single closest-hit rays of the persistent tracer.  The in-kernel walks (PTR_NEE_CLOSEST=0) are the literal form of the same
loops: both must give the same film, bit for bit, and the same ray counts."""
import os

import numpy as np
import pytest

import pathtracer_amd as pa

import common

pytestmark = pytest.mark.gpu
W, H = 64, 48


def render(name, closest, integ=pa.LI_VOLPATH, spp=6):
    old = os.environ.get("PTR_NEE_CLOSEST")
    os.environ["PTR_NEE_CLOSEST"] = "1" if closest else "0"
    try:
        sc = common.build(name, W, H)
        r = pa.Renderer(0)  # the switch is read when the context is created
    finally:
        if old is None:
            os.environ.pop("PTR_NEE_CLOSEST", None)
        else:
            os.environ["PTR_NEE_CLOSEST"] = old
    r.upload(sc)
    r.set_camera(sc.camera)
    r.film_init(W, H, 21)
    r.counters_reset()
    r.render(spp, integ)
    we, acc, smp = r.film()
    c = r.counters()
    prof_families = None
    r.close()
    sc.close()
    return we.copy(), c


@pytest.mark.parametrize("name", ["cornell", "cornell_glass", "cornell_fog", "cornell_disney", "room", "volcaustic", "zoo", "zoo_sky_fog"])
def test_closest_hit_queries_equal_the_walks(cuda, name):
    a, ca = render(name, True)
    b, cb = render(name, False)
    assert np.isfinite(a).all() and a.max() > 0
    assert np.array_equal(a, b)
    for k in ("total_paths", "shadow_rays", "mis_rays", "surface_hits", "medium_hits", "misses", "zero_radiance"):
        assert ca[k] == cb[k], k


@pytest.mark.parametrize("name", ["zoo", "room", "subsurface"])
def test_material_binned_shade_queue_gives_the_same_film(cuda, name, monkeypatch):
    """PTR_BIN_HITS=1 (k_miss compacts the vertices to shade into light and heavy materials, k_hitq_join joins the lists):
    another order of the same vertices, the same film bit for bit"""
    import numpy as np

    films = []
    for binned in ("0", "1"):
        monkeypatch.setenv("PTR_BIN_HITS", binned)  # read when the context is created
        sc = common.build(name, 96, 64)
        with pa.Renderer(0) as r:
            r.upload(sc)
            r.set_camera(sc.camera)
            for integ in (pa.LI_PATHSAMPLED, pa.LI_VOLPATH):
                r.film_init(96, 64, 4)
                r.render(6, integ)
                films.append(r.film()[0].copy())
        sc.close()
    assert np.array_equal(films[0], films[2]) and np.array_equal(films[1], films[3])
