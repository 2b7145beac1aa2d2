"""Scene descriptions on the Python side of the boundary.

Two sources of a ``PtrFlatScene``:

* :class:`BuiltScene` - one of the reference's scene families (src/main.cpp:459-1090), built by
  the C++ mirror of the reference's scene API in lib/libpt_host.so;
* :class:`FlatSceneBuilder` - small scenes assembled directly from numpy arrays (tests, tools).
"""
import ctypes as C
import os

import numpy as np

from . import capi


class BuiltScene:
    """name: cornell | knot | zoo | volcaustic | fluid | subsurface; p0 = size, p1 = variant."""

    def __init__(self, name, width, height, p0=0, p1=0, sky_exr=None):
        h = capi.host()
        h.pth_scene_sky_exr(os.fspath(sky_exr).encode() if sky_exr else None)  # zoo: the sky comes from this .exr (InsertEXRLightMap)
        try:
            self._h = h.pth_scene_build(name.encode(), int(width), int(height), int(p0), int(p1))
        finally:
            h.pth_scene_sky_exr(None)
        if not self._h:
            raise ValueError("unknown scene family %r" % name)
        self.name, self.width, self.height = name, int(width), int(height)
        self.flat = h.pth_scene_flat(self._h)  # POINTER(PtrFlatScene), owned by the handle
        self.camera = capi.PtrCamera()
        h.pth_scene_camera(self._h, C.byref(self.camera))
        self.rand_calls = h.pth_scene_rand_calls(self._h)

    def close(self):
        if self._h:
            capi.host().pth_scene_free(self._h)
            self._h = None
            self.flat = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def transform(m=None):
    """PtrTransform from a 4x4 matrix (numpy, row-major)."""
    t = capi.PtrTransform()
    m64 = np.eye(4) if m is None else np.asarray(m, dtype=np.float64)
    m32 = m64.astype(np.float32)
    inv = np.linalg.inv(m32.astype(np.float64)).astype(np.float32)
    if np.allclose(m64[3], [0, 0, 0, 1]):
        inv[3] = [0, 0, 0, 1]
    t.m[:] = m32.reshape(-1).tolist()
    t.mInv[:] = inv.reshape(-1).tolist()
    return t


def translate(x, y, z):
    m = np.eye(4)
    m[:3, 3] = [x, y, z]
    return m


def scale(x, y, z):
    return np.diag([x, y, z, 1.0])


def rotate_y(deg):
    a = np.radians(deg)
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]])


def rotate_x(deg):
    a = np.radians(deg)
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]])


class ImageTexture:
    """A spectrum slot that reads image `index` of the scene (FlatSceneBuilder.image) at the hit's uv."""

    def __init__(self, index):
        self.index = int(index)


def _tex_spec(t, rgb):
    if isinstance(rgb, ImageTexture):
        t.type = 2
        t.image = rgb.index
        return
    t.type = 0
    t.spec[:] = [float(v) for v in rgb]


def _tex_float(t, v):
    t.type = 1
    t.fval = float(v)


def material(kind, **kw):
    """Material descriptors in the slot order MakeMaterial reads (scene.cpp:801-889)."""
    m = capi.PtrMaterial()
    tx = m.textures
    if kind == "emissive":
        m.type = capi.MAT_NONE
        m.is_emissive = 1
        _tex_spec(tx[0], kw["L"])
        m.n_textures = 1
    elif kind == "matte":
        m.type = capi.MAT_MATTE
        _tex_spec(tx[0], kw.get("kd", (0.5, 0.5, 0.5)))
        _tex_float(tx[1], kw.get("sigma", 0.0))
        m.n_textures = 2
    elif kind == "mirror":
        m.type = capi.MAT_MIRROR
        _tex_spec(tx[0], kw.get("kr", (0.9, 0.9, 0.9)))
        m.n_textures = 1
    elif kind == "glass":
        m.type = capi.MAT_GLASS
        _tex_spec(tx[0], kw.get("kr", (1, 1, 1)))
        _tex_spec(tx[1], kw.get("kt", (1, 1, 1)))
        _tex_float(tx[2], kw.get("uroughness", 0.0))
        _tex_float(tx[3], kw.get("vroughness", 0.0))
        _tex_float(tx[4], kw.get("index", 1.5))
        m.n_textures = 5
    elif kind == "metal":
        m.type = capi.MAT_METAL
        _tex_spec(tx[0], kw.get("eta", (0.2, 0.92, 1.1)))
        _tex_spec(tx[1], kw.get("k", (3.9, 2.45, 2.14)))
        _tex_float(tx[2], kw.get("uroughness", 0.05))
        _tex_float(tx[3], kw.get("vroughness", 0.05))
        m.n_textures = 4
    elif kind == "plastic":
        m.type = capi.MAT_PLASTIC
        _tex_spec(tx[0], kw.get("kd", (0.25, 0.25, 0.25)))
        _tex_spec(tx[1], kw.get("ks", (0.25, 0.25, 0.25)))
        _tex_float(tx[2], kw.get("roughness", 0.1))
        m.n_textures = 3
    elif kind == "uber":
        m.type = capi.MAT_UBER
        _tex_spec(tx[0], kw.get("kd", (0.25, 0.25, 0.25)))
        _tex_spec(tx[1], kw.get("ks", (0.25, 0.25, 0.25)))
        _tex_spec(tx[2], kw.get("kr", (0, 0, 0)))
        _tex_spec(tx[3], kw.get("kt", (0, 0, 0)))
        _tex_float(tx[4], kw.get("uroughness", 0.1))
        _tex_float(tx[5], kw.get("vroughness", 0.1))
        _tex_spec(tx[6], kw.get("opacity", (1, 1, 1)))
        _tex_float(tx[7], kw.get("eta", 1.5))
        m.n_textures = 8
    elif kind == "disney":  # PTR_MAT_DISNEY: the reference's five Disney lobes (include/ptr_api.h)
        m.type = capi.MAT_DISNEY
        _tex_spec(tx[0], kw.get("color", (0.5, 0.5, 0.5)))
        _tex_float(tx[1], kw.get("roughness", 0.5))
        _tex_float(tx[2], kw.get("sheen", 0.0))
        _tex_float(tx[3], kw.get("sheen_tint", 0.5))
        _tex_float(tx[4], kw.get("clearcoat", 0.0))
        _tex_float(tx[5], kw.get("clearcoat_gloss", 1.0))
        _tex_float(tx[6], kw.get("flatness", 0.0))
        m.n_textures = 7
    elif kind == "subsurface":  # MakeSubsurfaceMaterial (scene.cpp): kr, kt, sigma_a, sigma_s (1/mm, times scale), roughness; vals scale, eta, g
        m.type = capi.MAT_SUBSURFACE
        _tex_spec(tx[0], kw.get("kr", (1, 1, 1)))
        _tex_spec(tx[1], kw.get("kt", (1, 1, 1)))
        _tex_spec(tx[2], kw.get("sigma_a", (0.0011, 0.0024, 0.014)))
        _tex_spec(tx[3], kw.get("sigma_s", (2.55, 3.21, 3.77)))
        _tex_float(tx[4], kw.get("uroughness", 0.0))
        _tex_float(tx[5], kw.get("vroughness", 0.0))
        m.vals[0], m.vals[1], m.vals[2] = float(kw.get("scale", 1.0)), float(kw.get("eta", 1.33)), float(kw.get("g", 0.0))
        m.n_textures = 6
    elif kind == "kdsubsurface":  # MakeKdSubsurfaceMaterial: kd, kt, kr, mean free path, roughness; vals eta, scale
        m.type = capi.MAT_KDSUBSURFACE
        _tex_spec(tx[0], kw.get("kd", (0.5, 0.5, 0.5)))
        _tex_spec(tx[1], kw.get("kt", (1, 1, 1)))
        _tex_spec(tx[2], kw.get("kr", (1, 1, 1)))
        _tex_spec(tx[3], kw.get("mfp", (1, 1, 1)))
        _tex_float(tx[4], kw.get("uroughness", 0.0))
        _tex_float(tx[5], kw.get("vroughness", 0.0))
        m.vals[0], m.vals[1] = float(kw.get("eta", 1.33)), float(kw.get("scale", 1.0))
        m.n_textures = 6
    elif kind == "translucent":
        m.type = capi.MAT_TRANSLUCENT
        _tex_spec(tx[0], kw.get("kd", (0.25, 0.25, 0.25)))
        _tex_spec(tx[1], kw.get("ks", (0.25, 0.25, 0.25)))
        _tex_spec(tx[2], kw.get("reflect", (0.5, 0.5, 0.5)))
        _tex_spec(tx[3], kw.get("transmit", (0.5, 0.5, 0.5)))
        _tex_float(tx[4], kw.get("roughness", 0.1))
        m.n_textures = 5
    else:
        raise ValueError(kind)
    return m


class _EnvTables:
    """keeps the tables of ptr_envmap_build alive as long as the scene that points at them"""

    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            capi.hip().ptr_envmap_free(self.handle)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class FlatSceneBuilder:
    def __init__(self):
        self.prims, self.meshes, self.clouds, self.images = [], [], [], []
        self._keep = []
        self.camera_medium = None

    def _prim(self, shape, params, mat, to_world=None, geom=0, medium=None, no_mat=False, reverse=False):
        p = capi.PtrPrimitive()
        p.shape_type = shape
        p.reverse_orientation = int(reverse)
        p.geom = geom
        p.no_mat = int(no_mat)
        p.to_world = transform(to_world)
        for i, v in enumerate(params):
            p.params[i] = float(v)
        if mat is not None:
            p.mat = mat
        if medium is not None:
            p.medium = medium
        self.prims.append(p)
        return len(self.prims) - 1

    def sphere(self, radius, mat, to_world=None, **kw):
        return self._prim(capi.SHAPE_SPHERE, [radius], mat, to_world, **kw)

    def rectangle(self, sx, sy, mat, to_world=None, **kw):
        return self._prim(capi.SHAPE_RECTANGLE, [sx, sy], mat, to_world, **kw)

    def disk(self, height, radius, inner, phi_max, mat, to_world=None, **kw):
        return self._prim(capi.SHAPE_DISK, [height, radius, inner, phi_max], mat, to_world, **kw)

    def box(self, sx, sy, sz, mat, to_world=None, **kw):
        return self._prim(capi.SHAPE_BOX_PROCEDURAL, [sx, sy, sz], mat, to_world, **kw)

    def sphere_procedural(self, radius, mat, to_world=None, **kw):
        return self._prim(capi.SHAPE_SPHERE_PROCEDURAL, [radius], mat, to_world, **kw)

    def mesh(self, vertices, triangles, mat, to_world=None, uvs=None, uv_triangles=None, **kw):
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        idx = np.zeros((t.shape[0], 3, 3), dtype=np.int32)
        idx[:, :, 0] = t
        m = capi.PtrMesh()
        m.n_triangles, m.n_vertices = t.shape[0], v.shape[0]
        if uvs is not None:
            uv = np.ascontiguousarray(uvs, dtype=np.float32).reshape(-1, 2)
            idx[:, :, 2] = t if uv_triangles is None else np.asarray(uv_triangles, dtype=np.int32).reshape(-1, 3)
            m.n_uvs = uv.shape[0]
            m.uv = uv.ctypes.data_as(capi.fptr)
            self._keep.append(uv)
        idx = np.ascontiguousarray(idx.reshape(-1))
        m.p = v.ctypes.data_as(capi.fptr)
        m.indices = idx.ctypes.data_as(capi.iptr)
        m.to_world = transform(to_world)
        self._keep += [v, idx]
        self.meshes.append(m)
        return self._prim(capi.SHAPE_MESH, [], mat, None, geom=len(self.meshes) - 1, **kw)

    def parsed_mesh(self, mesh, mat, to_world=None, **kw):
        """A sub-mesh as assets.load_obj returns it (ParsedMesh: per-corner position / normal / uv indices)."""
        v = np.ascontiguousarray(mesh["p"], dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(mesh["indices"], dtype=np.int32).reshape(-1)
        if idx.size % 9 or v.shape[0] == 0:
            raise ValueError("mesh without triangles")
        m = capi.PtrMesh()
        m.n_triangles, m.n_vertices = idx.size // 9, v.shape[0]
        uv = np.ascontiguousarray(mesh.get("uv", np.zeros((0, 2))), dtype=np.float32).reshape(-1, 2)
        if uv.shape[0]:
            m.n_uvs = uv.shape[0]
            m.uv = uv.ctypes.data_as(capi.fptr)
            self._keep.append(uv)
        m.p = v.ctypes.data_as(capi.fptr)
        m.indices = idx.ctypes.data_as(capi.iptr)
        m.to_world = transform(to_world)
        self._keep += [v, idx]
        self.meshes.append(m)
        return self._prim(capi.SHAPE_MESH, [], mat, None, geom=len(self.meshes) - 1, **kw)

    def image(self, rgb):
        """float32 [h, w, 3] texels, rows top-down (what LoadTextureImageData returns) -> ImageTexture for a spectrum slot"""
        a = np.ascontiguousarray(rgb, dtype=np.float32)
        im = capi.PtrImage()
        im.height, im.width = a.shape[0], a.shape[1]
        im.rgb = a.ctypes.data_as(capi.fptr)
        self._keep.append(a)
        self.images.append(im)
        return ImageTexture(len(self.images) - 1)

    def cloud(self, positions, radius, mat, **kw):
        p = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        c = capi.PtrCloud()
        c.n_particles, c.radius = p.shape[0], float(radius)
        c.positions = p.ctypes.data_as(capi.fptr)
        self._keep.append(p)
        self.clouds.append(c)
        return self._prim(capi.SHAPE_PARTICLECLOUD, [], mat, None, geom=len(self.clouds) - 1, **kw)

    def envmap(self, rgb, scale=(1, 1, 1), to_world=None, device=-1):
        """InsertLightMap for this scene: the light-map tables of float32 [h, w, 3] texels (rows top-down) built on the GPU
        (ptr_envmap_build: the floats the host pipeline gives), attached when the scene is finished."""
        a = np.ascontiguousarray(rgb, dtype=np.float32)
        sc3 = (C.c_float * 3)(*[float(v) for v in scale])
        h = C.c_void_p()
        capi.check(capi.hip().ptr_envmap_build(int(device), a.ctypes.data_as(capi.fptr), a.shape[1], a.shape[0], sc3, C.byref(h)))
        view = capi.PtrEnvMap()
        capi.check(capi.hip().ptr_envmap_view(h, C.byref(view), None))
        view.light_to_world = transform(to_world)
        self._env = (h, view)
        self._keep.append(_EnvTables(h))

    def distant_light(self, L, w):
        """Aggregator::InsertDistantLight(L, w) (primitive.cpp:180-189): radiance L from direction w (towards the light)"""
        d = capi.PtrDistantLight()
        d.L[:] = [float(v) for v in L]
        d.w[:] = [float(v) for v in w]
        if not hasattr(self, "_distant"):
            self._distant = []
        self._distant.append(d)

    def finish(self):
        fs = capi.PtrFlatScene()
        fs.n_primitives, fs.n_meshes, fs.n_clouds, fs.n_images = len(self.prims), len(self.meshes), len(self.clouds), len(self.images)
        pa = (capi.PtrPrimitive * max(1, len(self.prims)))(*self.prims)
        ma = (capi.PtrMesh * max(1, len(self.meshes)))(*self.meshes)
        ca = (capi.PtrCloud * max(1, len(self.clouds)))(*self.clouds)
        ia = (capi.PtrImage * max(1, len(self.images)))(*self.images)
        fs.primitives, fs.meshes, fs.clouds, fs.images = pa, ma, ca, ia
        if self.camera_medium is not None:
            fs.camera_medium = self.camera_medium
        if getattr(self, "_env", None) is not None:
            fs.has_envmap = 1
            fs.envmap = self._env[1]
        dl = getattr(self, "_distant", [])
        if dl:
            da = (capi.PtrDistantLight * len(dl))(*dl)
            fs.n_distant, fs.distant = len(dl), da
            self._keep.append(da)
        self._keep += [pa, ma, ca, ia]
        return PyScene(fs, self._keep)


def medium(sigma_a, sigma_s, g=0.0):
    m = capi.PtrMedium()
    m.sigma_a[:] = [float(v) for v in sigma_a]
    m.sigma_s[:] = [float(v) for v in sigma_s]
    m.g = float(g)
    m.is_valid = 1
    return m


class PyScene:
    """A flat scene whose arrays live in Python objects."""

    def __init__(self, fs, keep):
        self._fs, self._keep = fs, keep
        self.flat = C.pointer(fs)
        self.rand_calls = 0


def look_at_camera(eye, at, up, fov, aspect, aperture=0.0):
    """Camera::Config (camera.cpp:145-197) through the host mirror."""
    cam = capi.PtrCamera()
    arr = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    capi.host().pth_camera_config(arr(eye), arr(at), arr(up), float(fov), float(aspect), float(aperture), C.byref(cam))
    return cam
