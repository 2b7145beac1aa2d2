"""The generator of tests/test_gpu_random_scenes.py, on the CPU: its scenes build (ptr_scene_build accepts them), the oracle
renders them to finite, lit pictures under all four integrators, and a seed always gives the same scene.  (Seeds 64-71
build their sky tables on the GPU and are left to the GPU file.)"""
import ctypes as C

import numpy as np
import pytest

from oracle import orc
from pathtracer_amd import capi

import common
import test_gpu_random_scenes as T

SEEDS = [0, 7, 19, 33, 40, 45, 52, 55, 61, 73, 80]


@pytest.mark.parametrize("seed", SEEDS)
def test_random_scene_builds_and_renders_on_the_oracle(seed):
    sc, cam, eye = T.random_scene(seed)
    b, h1, h2 = C.c_void_p(), C.c_uint64(), C.c_uint64()
    libc = C.CDLL(None)
    libc.srand(1)  # the top-level tree draws its split axes from libc rand(), as Aggregator::Wrap does
    capi.check(capi.hip().ptr_scene_build(sc.flat, C.byref(b)))
    capi.check(capi.hip().ptr_scene_built_hash(b, C.byref(h1)))
    capi.check(capi.hip().ptr_scene_built_free(b))
    o = orc.OracleScene(sc.flat, 0)
    for integ in common.INTEGRATORS.values():
        we, acc, smp = o.render(cam, 32, 24, 3, 0, 2, integ, 4)
        img = orc.image_of(we, acc)
        assert np.isfinite(img).all() and img.max() > 0
    o.close()
    sc2, _, _ = T.random_scene(seed)  # the same seed is the same scene (a failure report names a seed)
    libc.srand(1)
    capi.check(capi.hip().ptr_scene_build(sc2.flat, C.byref(b)))
    capi.check(capi.hip().ptr_scene_built_hash(b, C.byref(h2)))
    capi.check(capi.hip().ptr_scene_built_free(b))
    assert h1.value == h2.value
