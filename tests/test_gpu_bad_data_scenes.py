"""Scenes whose *values* are garbage (NaN vertices, infinite positions, index of refraction 0, negative radii, absurd media,
unknown shape / material types) but whose indices are sound - the cases tests/test_cpu_scene_validation.py expects
ptr_scene_build to accept: the HIP path must trace and render them without a fault, a hang or a sticky error (pictures
may hold NaN: that is the data's doing), and must go on working afterwards."""
import numpy as np
import pytest

import pathtracer_amd as pa

import common
from test_cpu_scene_validation import ACCEPTED, valid_scene

pytestmark = pytest.mark.gpu
FMAX = np.finfo(np.float32).max


@pytest.mark.parametrize("corrupt", ACCEPTED, ids=[c.__name__.split("_", 1)[1] for c in ACCEPTED])
def test_garbage_values_do_not_fault(cuda, corrupt):
    sc = valid_scene()
    corrupt(sc._fs)
    cam = pa.look_at_camera((3, 2, 4), (0, 0, 0), (0, 1, 0), 45.0, 4 / 3)
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(cam)
        o, d = common.random_rays(2000, 1, -2.0, 2.0)
        g = common.gpu_trace_closest(r, o, d, np.full(2000, FMAX, np.float32))
        assert g["prim"].shape == (2000,)
        for integ in common.INTEGRATORS.values():
            r.film_init(32, 24, 3)
            r.render(4, integ)
            r.sync()
            assert r.image().shape == (24, 32, 3)
        # the context is still good for a sound scene
        r.upload(valid_scene())
        r.film_init(32, 24, 3)
        r.render(4, pa.LI_VOLPATH)
        r.sync()
        img = r.image()
        assert np.isfinite(img).all() and img.max() > 0
