"""The work distribution of the persistent tracer does not show in the film: sixteen fetch cursors per launch or one
(PTR_CURSORS), three passes in flight or one (PTR_LANES) - and films of a few pixels, whose launches carry fewer rays than
there are cursors, pools or lanes (1 x 1 pixel x 1 sample is one ray; the stretches of fifteen cursors are empty).  Scenes of
triangle meshes (the lean tracer build: leaf phase in pair mode), of analytic primitives and meshes (the four-wave build) and
of particle clouds (pair mode beside the item loop)."""
import os

import numpy as np
import pytest

import pathtracer_amd as pa

import common

pytestmark = pytest.mark.gpu
SEED = 5


def film_of(name, w, h, spp, integ, env):
    old = {k: os.environ.get(k) for k in ("PTR_CURSORS", "PTR_LANES")}
    for k in old:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        sc = common.build(name, w, h)
        with pa.Renderer(0) as r:  # the hooks are read by ptr_create
            r.upload(sc)
            r.set_camera(sc.camera)
            r.film_init(w, h, SEED)
            for _ in range(2):  # two calls: a second pass on the next lane, or on the same one
                r.render(spp, integ)
            r.sync()
            out = [a.copy() for a in r.film()]
        sc.close()
        return out
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("name", ["knot_glass", "room", "fluid"])
@pytest.mark.parametrize("size", [(1, 1, 1), (3, 5, 2), (17, 9, 5), (64, 48, 7)])
def test_film_does_not_depend_on_cursors_or_lanes(cuda, name, size):
    w, h, spp = size
    want = film_of(name, w, h, spp, pa.LI_VOLPATH, {"PTR_CURSORS": "1", "PTR_LANES": "1"})
    for env in ({}, {"PTR_CURSORS": "4"}, {"PTR_LANES": "6"}):
        got = film_of(name, w, h, spp, pa.LI_VOLPATH, env)
        for a, b in zip(want, got):
            assert np.array_equal(a, b), (name, size, env)
    assert int(want[2].sum()) == 2 * spp * w * h  # (samples per pixel: every sample arrived)
