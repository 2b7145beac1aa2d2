"""The HIP path against the closed forms of tests/test_cpu_known_answers.py: floor under a disk / sphere emitter,
emitter seen through an absorbing ball."""
import numpy as np
import pytest

import pathtracer_amd as pa
from oracle import orc

from test_cpu_known_answers import (H, INTEGRATORS, SEED, W, central, check_light, chord_transmittance, light_scene,
                                    medium_scene)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["disk", "sphere"])
def test_hip_floor_under_an_area_light(cuda, kind):
    sc, cam, want = light_scene(kind)
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(cam)
        for name, integ in INTEGRATORS.items():
            r.film_init(W, H, SEED)
            r.render(1024, integ)
            check_light(kind, name, central(r.image()).mean(), want)


def test_hip_transmittance_through_an_absorbing_ball(cuda):
    sc, cam, sa, rad = medium_scene()
    o = orc.OracleScene(sc.flat, 0)
    want = chord_transmittance(o, cam, sa, rad)  # geometry only: camera rays and the chord through the ball
    o.close()
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(cam)
        r.film_init(W, H, SEED)
        r.render(1024, pa.LI_VOLPATH)
        img = r.image()
    got, exp = central(img).mean(), central(want).mean()
    assert abs(got - exp) <= 0.015 * exp, (got, exp)
    assert np.allclose(img[0, 0], 1.0, atol=1e-5)
