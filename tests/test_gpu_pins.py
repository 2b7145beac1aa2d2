"""tests/pins.py held against the HIP path (ptr_unit_eval / ptr_unit_bsdf through the C-ABI), and the device values
against the oracle's on the same inputs."""
import numpy as np
import pytest

import pathtracer_amd as pa
from oracle import orc

import pins

pytestmark = pytest.mark.gpu


class GpuBackend:
    def __init__(self, scene):
        self.scene = scene
        self.r = pa.Renderer(0)
        self.r.upload(scene)

    def ev(self, kind, rows, prim=-1):
        return self.r.unit_eval(kind, rows, prim)

    def bsdf_eval(self, prim, wo, wi):
        return self.r.unit_bsdf(prim, wo, wi=wi)

    def bsdf_sample(self, prim, wo, u2):
        return self.r.unit_bsdf(prim, wo, u2=u2)


@pytest.fixture(scope="module")
def zoo(cuda):
    b = GpuBackend(pa.BuiltScene("zoo", 32, 24, 0, 1))
    yield b
    b.r.close()


@pytest.fixture(scope="module")
def mats(cuda):
    b = GpuBackend(pins.material_scene())
    yield b
    b.r.close()


@pytest.fixture(scope="module")
def sss(cuda):
    b = GpuBackend(pa.BuiltScene("subsurface", 32, 24, 0, 0))
    yield b
    b.r.close()


def test_fresnel(zoo):
    pins.check_fresnel(zoo)


def test_ggx(zoo):
    pins.check_ggx(zoo)


def test_henyey_greenstein(zoo):
    pins.check_hg(zoo)


def test_light_map_distribution(zoo):
    e = zoo.scene.flat.contents.envmap
    pins.check_env(zoo, e.nu, e.nv)


@pytest.mark.parametrize("name", ["matte", "oren_nayar", "plastic", "metal", "rough_glass", "translucent_diffuse", "translucent_glossy",
                                  "disney_diffuse", "disney_full", "disney_coat"])
def test_material_integrals(mats, name):
    pins.check_material_integrals(mats, name, pins.MATS[name], fine=name in ("metal", "disney_full", "disney_coat"))


@pytest.mark.parametrize("name", ["disney_diffuse", "disney_full", "disney_coat"])
def test_disney_lobes_against_the_published_formulas(mats, name):
    pins.check_disney_closed_form(mats, name, pins.MATS[name])


@pytest.mark.parametrize("name", ["matte", "oren_nayar", "plastic", "metal", "disney_diffuse", "disney_full", "disney_coat"])
def test_reciprocity(mats, name):
    pins.check_reciprocity(mats, name, pins.MATS[name])


def test_specular_lobes(mats):
    pins.check_specular(mats, pins.MATS["mirror"], pins.MATS["glass"])


@pytest.mark.parametrize("prim", [1, 3])
def test_bssrdf_profile(sss, prim):
    pins.check_bssrdf(sss, prim)


def test_device_values_match_the_oracle(zoo):
    """the same records through both: equal up to the device's libm (a few ulp)"""
    o = orc.OracleScene(zoo.scene.flat, zoo.scene.rand_calls)
    rng = np.random.default_rng(11)
    n = 20000
    wo = rng.normal(size=(n, 3)); wo[:, 2] = np.abs(wo[:, 2]) + 0.02
    wo = (wo / np.linalg.norm(wo, axis=1, keepdims=True)).astype(np.float32)
    wi = rng.normal(size=(n, 3))
    wi = (wi / np.linalg.norm(wi, axis=1, keepdims=True)).astype(np.float32)
    u = rng.random((n, 2)).astype(np.float32)
    for name, prim in pins.ZOO.items():
        f, p = zoo.bsdf_eval(prim, wo, wi)
        f0, p0 = orc.bsdf_eval(o, prim, wo, wi)
        assert np.allclose(f, f0, rtol=2e-4, atol=1e-6), name
        assert np.allclose(p, p0, rtol=2e-4, atol=1e-6), name
        swi, sf, sp, fl = zoo.bsdf_sample(prim, wo, u)
        swi0, sf0, sp0, fl0 = orc.bsdf_sample(o, prim, wo, u)
        same = fl == fl0
        assert same.mean() > 0.999, name  # a lobe choice can flip where u sits on a boundary
        assert np.allclose(swi[same], swi0[same], atol=2e-4), name
        assert np.allclose(sp[same], sp0[same], rtol=3e-3, atol=1e-5), name
    for kind, rows in (("ggx_d", np.concatenate([np.full((n, 2), [0.2, 0.35], np.float32), np.abs(wi)], 1)),
                       ("ggx_sample_wh", np.concatenate([np.full((n, 2), [0.2, 0.35], np.float32), wo, u], 1)),
                       ("hg_sample", np.concatenate([np.full((n, 1), 0.6, np.float32), wo, u], 1)),
                       ("env_sample", u)):
        a, b = zoo.ev(kind, rows), orc.unit_eval(kind, rows, o)
        assert np.allclose(a, b, rtol=1e-3, atol=2e-4), kind
    o.close()
