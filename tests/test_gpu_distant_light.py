"""The HIP path under distant lights: the closed forms of tests/test_cpu_distant_light.py, and images of a scene lit by a distant
and an area light together against the oracle under all four integrators."""
import numpy as np
import pytest

import pathtracer_amd as pa
from oracle import orc

import common
from test_cpu_distant_light import H, INTEGRATORS, SEED, W, check_floor, check_shadow, floor_scene

pytestmark = pytest.mark.gpu


def test_hip_floor_under_a_distant_light(cuda):
    sc, cam = floor_scene()
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(cam)
        for name, integ in INTEGRATORS.items():
            r.film_init(W, H, SEED)
            r.render(4, integ)
            check_floor(name, r.image())


def test_hip_shadow_of_a_distant_light(cuda):
    sc, cam = floor_scene(occluder=True)
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(cam)
        r.film_init(W, H, SEED)
        r.render(4, pa.LI_DIRECT)
        img = r.image()
    check_shadow(img)


@pytest.mark.parametrize("name", list(INTEGRATORS))
def test_hip_distant_and_area_light_against_the_oracle(cuda, name):
    sc, cam = floor_scene(extra=True)
    spp = 8
    o = orc.OracleScene(sc.flat, 0)
    we, acc, _ = o.render(cam, W, H, SEED, 0, spp, INTEGRATORS[name], 8)
    ref = orc.image_of(we, acc).reshape(H, W, 3)
    oc = o.counters() if hasattr(o, "counters") else None
    o.close()
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(cam)
        r.film_init(W, H, SEED)
        r.render(spp, INTEGRATORS[name])
        img = r.image()
    mean_rel, rmse_rel, off = common.image_stats(img, ref)
    assert mean_rel < 5e-3 and rmse_rel < 5e-2 and off < 0.02, (name, mean_rel, rmse_rel, off)
    if name != "path":
        assert ref.mean() > 0.01
