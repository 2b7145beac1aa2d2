"""tests/pins.py held against the CPU oracle: what pins the otherwise unpinned restatement."""
import numpy as np
import pytest

import pathtracer_amd as pa
from oracle import orc

import pins


class OracleBackend:
    def __init__(self, scene):
        self.scene = scene
        self.o = orc.OracleScene(scene.flat, scene.rand_calls)

    def ev(self, kind, rows, prim=-1):
        return orc.unit_eval(kind, rows, self.o, prim)

    def bsdf_eval(self, prim, wo, wi):
        return orc.bsdf_eval(self.o, prim, wo, wi)

    def bsdf_sample(self, prim, wo, u2):
        return orc.bsdf_sample(self.o, prim, wo, u2)


@pytest.fixture(scope="module")
def zoo():
    return OracleBackend(pa.BuiltScene("zoo", 32, 24, 0, 1))  # variant 1: with the light-map sky


@pytest.fixture(scope="module")
def mats():
    return OracleBackend(pins.material_scene())


@pytest.fixture(scope="module")
def sss():
    return OracleBackend(pa.BuiltScene("subsurface", 32, 24, 0, 0))


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


@pytest.mark.parametrize("prim", [1, 3])  # SubsurfaceScene: 1 = tabulated "Ketchup" coefficients, 3 = KdSubsurface (inverted albedo)
def test_bssrdf_profile(sss, prim):
    pins.check_bssrdf(sss, prim)
