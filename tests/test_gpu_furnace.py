"""The HIP path against the closed-form answers of the white furnace (see tests/test_cpu_furnace.py): albedo,
reflectance and 1 - numbers that no implementation had a hand in."""
import numpy as np
import pytest

import pathtracer_amd as pa

from test_cpu_furnace import CASES, INTEGRATORS, W, H, SEED, body_mask, check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(CASES))
def test_hip_white_furnace(cuda, name):
    variant, want, integrators, tol = CASES[name]
    sc, inner = body_mask(variant)
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(sc.camera)
        for integ in integrators:
            r.film_init(W, H, SEED)
            r.render(256, INTEGRATORS[integ])  # 5x the oracle's samples: the GPU has them to spare
            img = r.image()
            check(img, inner, want, tol / 2 if want < 1 else tol, (name, integ))
            if not inner[0, 0]:
                assert np.allclose(img[0, 0], 1.0, atol=1e-5)


def test_hip_rough_glass_energy(cuda):
    sc, inner = body_mask(3)
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(sc.camera)
        r.film_init(W, H, SEED)
        r.render(256, pa.LI_VOLPATH)
        m = r.image()[inner].mean()
    assert 0.5 < m < 1.005, m  # alpha = RoughnessToAlpha(0.1) = 0.46: two rough interfaces without a multiple-scattering term lose a third
