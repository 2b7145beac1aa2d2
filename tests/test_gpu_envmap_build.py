"""SURVEY 8f row 4, device half: the light-map tables (MipMap level 0 after scaling / Lanczos resampling, the
sin-weighted luminance table, row CDFs, row integrals, marginal CDF) built by ptr_envmap_build on the GPU against the
ORACLE's restatement of texture.cpp:111-260 / infinite.cpp:7-39 (oracle/orc_envmap.cpp; pinned by closed forms in
tests/test_cpu_envmap_oracle.py) - the same floats, bit for bit, for power-of-two and other sizes - and, as a
consequence, against the host mirror's InsertLightMap, which the CPU suite holds to the same oracle."""
import ctypes as C

import numpy as np
import pytest

import pathtracer_amd as pa
from pathtracer_amd import capi
from oracle import orc

pytestmark = pytest.mark.gpu


def tables_of(rgb, scale, on_device):
    h, w, _ = rgb.shape
    flat = np.ascontiguousarray(rgb, dtype=np.float32)
    sc3 = (C.c_float * 3)(*scale)
    handle = capi.host().pth_lightmap_scene(flat.ctypes.data_as(C.POINTER(C.c_float)), w, h, sc3, on_device)
    assert handle
    fs = capi.host().pth_scene_flat(handle).contents
    assert fs.has_envmap
    e = fs.envmap
    out = {"dims": (e.width, e.height, e.nu, e.nv),
           "texels": np.ctypeslib.as_array(e.texels, (3 * e.width * e.height,)).copy(),
           "func": np.ctypeslib.as_array(e.func, (e.nu * e.nv,)).copy(),
           "cdf": np.ctypeslib.as_array(e.cdf, ((e.nu + 1) * e.nv,)).copy(),
           "func_int": np.ctypeslib.as_array(e.func_int, (e.nv,)).copy(),
           "marg_cdf": np.ctypeslib.as_array(e.marg_cdf, (e.nv + 1,)).copy(),
           "marg_int": np.float32(e.marg_int)}
    capi.host().pth_scene_free(handle)
    return out


def sky(w, h, seed, hdr=False):
    rng = np.random.default_rng(seed)
    img = rng.uniform(0.0, 1.0, (h, w, 3)).astype(np.float32)
    if hdr:  # a sun: values far above 1 (kept for power-of-two maps, clamped by the reference's resample path otherwise)
        img[h // 3, w // 4] = (900.0, 800.0, 500.0)
    img[h - 2:] = 0.0  # rows of zero radiance: funcInt = 0 takes Distribution1D's uniform branch
    return img


@pytest.mark.parametrize("w,h,scale,hdr", [
    (512, 256, (2.5, 2.5, 2.5), True),       # the knot scene's sky size: no resample, 9-level pyramid
    (16, 8, (1.0, 1.0, 1.0), False),
    (64, 64, (1.0, 0.5, 0.25), True),
    (100, 37, (1.0, 1.0, 1.0), False),      # both axes resampled (128 x 64)
    (640, 256, (2.0, 1.0, 3.0), True),      # one axis resampled
    (1000, 500, (1.0, 1.0, 1.0), True),     # -> 1024 x 512, tables of 2048 x 1024
    (3, 2, (1.0, 1.0, 1.0), False),
    (1, 1, (4.0, 4.0, 4.0), False),         # a single texel: the lookup returns the last level
])
def test_device_tables_equal_host_tables(cuda, w, h, scale, hdr):
    img = sky(w, h, 7 + w, hdr)
    host = tables_of(img, scale, 0)
    dev = tables_of(img, scale, 1)
    ref = orc.envmap_tables(img, scale)  # the checker: CPU restatement of the reference
    assert ref["dims"] == dev["dims"]
    for k in ("texels", "func", "cdf", "func_int", "marg_cdf"):
        assert ref[k].shape == dev[k].shape, k
        assert np.array_equal(ref[k].view(np.uint32), dev[k].view(np.uint32)), k
    assert ref["marg_int"].view(np.uint32) == dev["marg_int"].view(np.uint32)
    assert host["dims"] == dev["dims"]
    for k in ("texels", "func", "cdf", "func_int", "marg_cdf"):
        assert host[k].shape == dev[k].shape, k
        assert np.array_equal(host[k].view(np.uint32), dev[k].view(np.uint32)), k
    assert host["marg_int"].view(np.uint32) == dev["marg_int"].view(np.uint32)
    # and they are a distribution: rows end at 1, the marginal too
    nu, nv = host["dims"][2], host["dims"][3]
    cdf = dev["cdf"].reshape(nv, nu + 1)
    assert (cdf[:, 0] == 0).all() and np.allclose(cdf[:, -1], 1, atol=1e-5) and (np.diff(cdf, axis=1) >= 0).all()
    assert dev["marg_cdf"][0] == 0 and abs(dev["marg_cdf"][-1] - 1) < 1e-5


def test_scene_with_device_built_sky_renders_the_same(cuda):
    """A scene family whose sky goes through InsertLightMap, tables from the device: the film is the film of the host tables."""
    films = []
    for mode in (0, 1):
        capi.host().SetLightMapBuilderC(mode)
        try:
            sc = pa.BuiltScene("zoo", 96, 64, 0, 0)
        finally:
            capi.host().SetLightMapBuilderC(0)
        r = pa.Renderer(0)
        r.upload(sc)
        r.set_camera(sc.camera)
        r.film_init(96, 64, 3)
        r.render(4, pa.LI_VOLPATH)
        films.append(r.film()[0].copy())
        r.close()
        sc.close()
    assert np.array_equal(films[0], films[1])


def test_bad_arguments(cuda):
    out = C.c_void_p()
    sc3 = (C.c_float * 3)(1, 1, 1)
    px = (C.c_float * 3)(1, 1, 1)
    assert capi.hip().ptr_envmap_build(-1, px, 0, 1, sc3, C.byref(out)) == capi.PTR_ERR_INVALID
    assert capi.hip().ptr_envmap_build(-1, None, 1, 1, sc3, C.byref(out)) == capi.PTR_ERR_INVALID
