"""Asset ingest through the host mirror (host/pt_loaders.cpp): the reference's OBJ loader
(src/third/obj_loader.cpp:261-443), MTL parser + MakeMTLMaterial (src/third/mtl.cpp,
src/core/scene.cpp:484-577), particle parser (src/third/part_loader.cpp:55-130) and texture image
loader (src/core/texture.cpp:29-64), returning numpy copies that FlatSceneBuilder takes as is."""
import ctypes as C
import os

import numpy as np

from . import capi

MATERIAL_TYPES = {1: "matte", 2: "mirror", 3: "glass", 4: "metal", 5: "translucent", 6: "plastic", 7: "uber",
                  8: "subsurface", 9: "kdsubsurface", 10: "playground"}


def _text(fn, *args):
    n = fn(*args, None, 0)
    buf = C.create_string_buffer(n + 1)
    fn(*args, buf, n + 1)
    return buf.value.decode("utf-8", "replace")


def _copy(ptr, shape, dtype):
    n = int(np.prod(shape))
    if not ptr or n == 0:
        return np.zeros(shape, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True).reshape(shape)


def parse_double(text):
    """ParseDouble (mtl.cpp:41-154): (ok, float32 value)."""
    out = C.c_float(0)
    ok = capi.host().pth_parse_double(str(text).encode(), C.byref(out))
    return bool(ok), np.float32(out.value)


def load_obj(path, split_mesh=True):
    """LoadObj: list of sub-meshes, each a dict with
    p [nV,3] f32, n [nN,3] f32, uv [nU,2] f32, indices [nTri,3,3] i32 (corner -> position, normal, uv; -1 = absent),
    mtl_file, mtl_name.  A file that cannot be opened gives []."""
    H = capi.host()
    h = H.pth_obj_load(os.fspath(path).encode(), int(bool(split_mesh)))
    try:
        out = []
        for i in range(H.pth_obj_count(h)):
            counts = (C.c_int * 4)()
            p, n, uv, idx = capi.fptr(), capi.fptr(), capi.fptr(), capi.iptr()
            H.pth_obj_mesh(h, i, counts, C.byref(p), C.byref(n), C.byref(uv), C.byref(idx))
            nt, nv, nn, nu = counts
            out.append({
                "p": _copy(p, (nv, 3), np.float32), "n": _copy(n, (nn, 3), np.float32), "uv": _copy(uv, (nu, 2), np.float32),
                "indices": _copy(idx, (nt, 3, 3), np.int32),
                "mtl_file": _text(H.pth_obj_mtl, h, i, 0), "mtl_name": _text(H.pth_obj_mtl, h, i, 1),
            })
        return out
    finally:
        H.pth_obj_free(h)


def load_obj_only(path):
    """LoadObjOnly: the whole file as one mesh (None when it has no faces)."""
    m = load_obj(path, split_mesh=False)
    return m[0] if m else None


class MtlLibrary:
    """MTLParse over one file (`times` > 1 lists it that often, as MTLParseAll does per sub-mesh)."""

    def __init__(self, file, basedir="", times=1):
        self._h = capi.host().pth_mtl_load(str(file).encode(), str(basedir).encode(), int(times))
        if not self._h:
            raise OSError("could not open %s%s" % (basedir, file))

    def close(self):
        if self._h:
            capi.host().pth_mtl_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return capi.host().pth_mtl_count(self._h)

    def names(self):
        return [_text(capi.host().pth_mtl_name, self._h, i) for i in range(len(self))]

    def find(self, name):
        """MTLFindMaterial: index of the first material of that name, 0 when there is none."""
        return capi.host().pth_mtl_find(self._h, str(name).encode())

    def value(self, i, key):
        buf = C.create_string_buffer(4096)
        found = capi.host().pth_mtl_value(self._h, int(i), key.encode(), buf, 4096)
        return buf.value.decode("utf-8", "replace") if found else None

    def spectrum(self, i, key):
        rgb = (C.c_float * 3)()
        found = capi.host().pth_mtl_spectrum(self._h, int(i), key.encode(), rgb)
        return np.array(rgb[:], np.float32) if found else None

    def material_numbers(self, i):
        """MakeMTLMaterial as numbers: (type name or 'emissive', textures) with each texture
        ('spectrum', rgb) / ('float', v) / ('image', width)."""
        head = (C.c_int * 2)()
        tex = (C.c_float * 96)()
        capi.host().pth_mtl_material(self._h, int(i), head, tex)
        t = np.array(tex[:], np.float32).reshape(16, 6)
        kind = "emissive" if head[1] else MATERIAL_TYPES.get(head[0], str(head[0]))
        return kind, t

    def material(self, i):
        """The PtrMaterial FlatSceneBuilder takes (colour textures only; image maps go through the C++ mirror)."""
        from .scene import material
        kind, t = self.material_numbers(i)
        if kind == "emissive":
            return material("emissive", L=t[0, 1:4])
        if kind == "plastic":
            return material("plastic", kd=t[0, 1:4], ks=t[1, 1:4], roughness=float(t[2, 4]))
        if kind == "matte":
            kd = t[0, 1:4] if t[0, 0] == 0 else [t[0, 4]] * 3
            return material("matte", kd=kd, sigma=float(t[1, 4]))
        if kind == "uber":
            return material("uber", kd=t[0, 1:4], ks=t[1, 1:4], kr=t[2, 1:4], kt=t[3, 1:4], uroughness=float(t[4, 4]), vroughness=float(t[5, 4]),
                            opacity=t[6, 1:4], eta=float(t[7, 4]))
        raise ValueError(kind)


def load_image(path):
    """LoadTextureImageData: float32 [h, w, 3] in [0, 1], rows top-down; None when there is no decoder."""
    w, h, ptr = C.c_int(), C.c_int(), capi.fptr()
    hd = capi.host().pth_image_load(os.fspath(path).encode(), C.byref(w), C.byref(h), C.byref(ptr))
    if not hd:
        return None
    try:
        return _copy(ptr, (h.value, w.value, 3), np.float32)
    finally:
        capi.host().pth_image_free(hd)


def load_exr(path):
    """ReadImageEXR (image_util.cpp:21-45): float32 [h, w, 3], rows top-down; None when the file cannot be read."""
    w, h = C.c_int(), C.c_int()
    p = capi.host().pth_exr_read(os.fspath(path).encode(), C.byref(w), C.byref(h))
    if not p:
        return None
    try:
        return _copy(C.cast(p, capi.fptr), (h.value, w.value, 3), np.float32)
    finally:
        capi.host().pth_exr_free(p)


def load_particles(path, fmt="vs"):
    """PParserInitialize + PParserParse: ([vectors [n,3]...], [scalars [n]...]) in format order; None when the file
    is missing, has no header or its record count differs from the header."""
    H = capi.host()
    h = H.pth_particles_load(os.fspath(path).encode(), fmt.encode())
    if not h:
        return None
    try:
        vecs, scal = [], []
        for c in fmt:
            cnt = C.c_int()
            if c in "vV":
                p = H.pth_particles_get(h, 0, len(vecs), C.byref(cnt))
                vecs.append(_copy(p, (cnt.value, 3), np.float32))
            elif c in "sS":
                p = H.pth_particles_get(h, 1, len(scal), C.byref(cnt))
                scal.append(_copy(p, (cnt.value,), np.float32))
        return vecs, scal
    finally:
        H.pth_particles_free(h)
