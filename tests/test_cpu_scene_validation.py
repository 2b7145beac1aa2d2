"""What ptr_scene_build (the host half of ptr_scene_upload, include/ptr_api.h) does with malformed flat scenes.

Every index a kernel will follow comes out of this function, so a descriptor it accepts must be safe to traverse: the
GPU has no bounds checks, an out-of-range index there is a memory fault.  Each case corrupts one field of a valid scene
and expects the documented error code (or acceptance where the value is only data), never a crash.  No GPU needed: the
scene is built, inspected and freed on the host."""
import ctypes as C

import numpy as np
import pytest

import pathtracer_amd as pa
from pathtracer_amd import capi
from pathtracer_amd import scene as S

INVALID, LIMIT = capi.PTR_ERR_INVALID, capi.PTR_ERR_LIMIT


def valid_scene():
    b = pa.FlatSceneBuilder()
    b.rectangle(4, 4, pa.material("matte", kd=(0.5, 0.5, 0.5)), S.translate(0, -1, 0) @ S.rotate_x(90))
    v = np.array([[-1, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    t = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    b.mesh(v, t, pa.material("glass", index=1.5), S.translate(0, 0.2, 0), uvs=np.array([[0, 0], [1, 0], [0, 1], [1, 1]], np.float32))
    b.cloud(np.array([[0, 0, 0], [0.5, 0.5, 0.5]], np.float32), 0.1, pa.material("plastic"))
    b.sphere(0.3, pa.material("emissive", L=(5, 5, 5)), S.translate(0, 2, 0))
    b.sphere(0.5, None, S.translate(1, 0, 0), medium=pa.medium((0.1, 0.1, 0.1), (0.5, 0.5, 0.5), 0.3), no_mat=True)
    return b.finish()


def build(sc):
    h = C.c_void_p()
    rc = capi.hip().ptr_scene_build(sc.flat, C.byref(h))
    if rc == capi.PTR_OK:
        info = capi.PtrBvhBuildInfo()
        assert capi.hip().ptr_scene_built_info(h, C.byref(info)) == capi.PTR_OK
        assert capi.hip().ptr_scene_built_free(h) == capi.PTR_OK
    else:
        assert not h.value
        assert capi.hip().ptr_last_error()  # says what was wrong
    return rc


def test_the_valid_scene_builds():
    assert build(valid_scene()) == capi.PTR_OK


def corrupt_counts(fs):
    fs.n_meshes = -1


def corrupt_null_prims(fs):
    fs.n_primitives = 3
    fs.primitives = C.cast(None, C.POINTER(capi.PtrPrimitive))


def corrupt_vertex_index(fs):
    fs.meshes[0].indices[9] = 4  # n_vertices is 4


def corrupt_vertex_index_negative(fs):
    fs.meshes[0].indices[0] = -1


def corrupt_mesh_null(fs):
    fs.meshes[0].p = C.cast(None, capi.fptr)


def corrupt_mesh_count(fs):
    fs.meshes[0].n_triangles = -5


def corrupt_mesh_ref(fs):
    fs.primitives[1].geom = 1  # one mesh only


def corrupt_cloud_ref(fs):
    fs.primitives[2].geom = -1


def corrupt_cloud_null(fs):
    fs.clouds[0].positions = C.cast(None, capi.fptr)


def corrupt_cloud_count(fs):
    fs.clouds[0].n_particles = -1


def corrupt_projective(fs):
    fs.primitives[0].to_world.m[12] = 0.5


def corrupt_mesh_projective(fs):
    fs.meshes[0].to_world.mInv[15] = 2.0


def corrupt_texture_image(fs):
    t = fs.primitives[0].mat.textures[0]
    t.type = 2
    t.image = 0  # the scene has no images


def corrupt_material_type(fs):
    fs.primitives[0].mat.type = 99


def corrupt_material_type_negative(fs):
    fs.primitives[0].mat.type = -3


def corrupt_shape_type(fs):
    fs.primitives[0].shape_type = 42


REFUSED = [
    (corrupt_counts, INVALID), (corrupt_null_prims, INVALID), (corrupt_vertex_index, INVALID), (corrupt_vertex_index_negative, INVALID),
    (corrupt_mesh_null, INVALID), (corrupt_mesh_count, INVALID), (corrupt_mesh_ref, INVALID), (corrupt_cloud_ref, INVALID),
    (corrupt_cloud_null, INVALID), (corrupt_cloud_count, INVALID), (corrupt_projective, LIMIT), (corrupt_mesh_projective, LIMIT),
    (corrupt_texture_image, INVALID),
]


@pytest.mark.parametrize("corrupt,code", REFUSED, ids=[c.__name__[8:] for c, _ in REFUSED])
def test_malformed_scenes_are_refused(corrupt, code):
    sc = valid_scene()
    corrupt(sc._fs)
    assert build(sc) == code
    assert build(valid_scene()) == capi.PTR_OK  # and the library is none the worse for it


def data_nan_params(fs):
    fs.primitives[3].params[0] = float("nan")  # a sphere of radius NaN: bounds NaN, never hit


def data_nan_vertex(fs):
    fs.meshes[0].p[4] = float("nan")


def data_inf_position(fs):
    fs.clouds[0].positions[0] = float("inf")


def data_uv_index(fs):
    fs.meshes[0].indices[2] = 1000  # a uv index beyond the list: the lookup is skipped on the device (mesh.cpp's own range test)
    fs.meshes[0].indices[5] = -7


def data_zero_eta(fs):
    fs.primitives[1].mat.textures[4].fval = 0.0  # index of refraction 0: NaNs in the Fresnel terms, no indexing involved


def data_negative_radius(fs):
    fs.clouds[0].radius = -1.0


def data_huge_medium(fs):
    fs.primitives[4].medium.sigma_s[:] = [1e30, 1e30, 1e30]


# an unknown shape type is skipped with the reference's own message (scene.cpp:911-915); a material type outside the enum
# leaves the primitive without material, as the reference's MakeMaterial switch does
ACCEPTED = [corrupt_material_type, corrupt_material_type_negative, corrupt_shape_type, data_nan_params, data_nan_vertex, data_inf_position, data_uv_index, data_zero_eta, data_negative_radius, data_huge_medium]


@pytest.mark.parametrize("corrupt", ACCEPTED, ids=[c.__name__.split("_", 1)[1] for c in ACCEPTED])
def test_bad_values_that_are_only_data_are_accepted(corrupt):
    """NaN / infinite / absurd values in positions, radii, material parameters change pictures, not indices"""
    sc = valid_scene()
    corrupt(sc._fs)
    assert build(sc) == capi.PTR_OK


def test_random_field_corruption_never_crashes():
    """400 scenes with one to three integer or pointer-free fields overwritten by extreme values: built or refused"""
    rng = np.random.default_rng(5)
    extremes = [-1, -2 ** 31, 2 ** 31 - 1, 0, 1, 2, 7, 255, 65536]
    seen = {capi.PTR_OK: 0}
    for _ in range(400):
        sc = valid_scene()
        fs = sc._fs
        for _ in range(int(rng.integers(1, 4))):
            v = int(rng.choice(extremes))
            k = int(rng.integers(0, 12))
            p = fs.primitives[int(rng.integers(0, 5))]
            if k == 0: p.shape_type = v
            elif k == 1: p.geom = v
            elif k == 2: p.no_mat = v
            elif k == 3: p.reverse_orientation = v
            elif k == 4: p.mat.type = v
            elif k == 5: p.mat.is_emissive = v
            elif k == 6: p.mat.n_textures = v
            elif k == 7: p.mat.textures[int(rng.integers(0, 16))].type = v
            elif k == 8: p.mat.textures[int(rng.integers(0, 16))].image = v
            elif k == 9: p.medium.is_valid = v
            elif k == 10: fs.meshes[0].indices[int(rng.integers(0, 18))] = v
            else: fs.meshes[0].n_uvs = v if v >= 0 and v <= 4 else 4  # (n_uvs beyond the array would be the caller lying about its buffer)
        rc = build(sc)
        seen[rc] = seen.get(rc, 0) + 1
        assert rc in (capi.PTR_OK, INVALID, LIMIT), rc
    assert seen[capi.PTR_OK] > 20 and seen.get(INVALID, 0) > 20, seen
