"""ctypes binding of oracle/liborc.so - TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; nothing under
pathtracer_amd/ may import this module.  The oracle is a CPU restatement of the reference's
hot path (PARITY UNPINNED: the reference cannot be built here).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liborc.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError("oracle/liborc.so missing: run `make -C oracle`")
        L = C.CDLL(LIB)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_trace_closest.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 9
        L.orc_trace_occluded.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.orc_generate_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int] + [C.c_void_p] * 3
        L.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_counters_read.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_counters_reset.argtypes = [C.c_void_p]
        L.orc_bvh_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_glibc_rand.restype = C.c_int
        L.orc_glibc_rand.argtypes = [C.c_int]
        L.orc_rng_draw.restype = C.c_float
        L.orc_rng_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_bsdf_eval.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.orc_bsdf_sample.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7
        L.orc_unit_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_film_rgb8.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        L.orc_envmap_build.restype = C.c_void_p
        L.orc_envmap_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_envmap_view.argtypes = [C.c_void_p, C.c_void_p] + [C.c_void_p] * 6
        L.orc_envmap_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


COUNTER_NAMES = ["total_paths", "misses", "surface_hits", "medium_hits", "light_hits", "zero_radiance",
                 "shadow_rays", "mis_rays", "probe_rays", "camera_rays", "nodes_visited", "prims_tested"]


class OracleScene:
    """flat: POINTER(PtrFlatScene) (or anything ctypes can pass as a pointer)."""

    def __init__(self, flat, rand_calls_before=0, meshfix=1):
        self._h = lib().orc_scene_create(C.cast(flat, C.c_void_p), int(rand_calls_before), int(meshfix))
        if not self._h:
            raise RuntimeError("orc_scene_create failed")

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trace_closest(self, o, d, tmax, order_dep=True):
        """o, d: float32 [3, n] planes; returns dict of numpy arrays."""
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        tmax = np.ascontiguousarray(tmax, dtype=np.float32)
        n = tmax.shape[0]
        prim = np.empty(n, np.int32); face = np.empty(n, np.int32); t = np.empty(n, np.float32)
        b0 = np.empty(n, np.float32); b1 = np.empty(n, np.float32)
        od = np.zeros(n, np.int32)
        lib().orc_trace_closest(self._h, n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data, prim.ctypes.data, face.ctypes.data,
                                t.ctypes.data, b0.ctypes.data, b1.ctypes.data, od.ctypes.data if order_dep else None)
        return {"prim": prim, "face": face, "t": t, "b0": b0, "b1": b1, "order_dep": od}

    def trace_occluded(self, o, d, tmax):
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        tmax = np.ascontiguousarray(tmax, dtype=np.float32)
        n = tmax.shape[0]
        occ = np.empty(n, np.int32)
        lib().orc_trace_occluded(self._h, n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data, occ.ctypes.data)
        return occ

    def generate_rays(self, cam, w, h, seed, sample):
        n = w * h
        o = np.empty((3, n), np.float32); d = np.empty((3, n), np.float32); wt = np.empty(n, np.float32)
        lib().orc_generate_rays(self._h, C.addressof(cam), w, h, C.c_uint64(seed), sample, o.ctypes.data, d.ctypes.data, wt.ctypes.data)
        return o, d, wt

    def render(self, cam, w, h, seed, sample_begin, ns, integrator, nthreads=8, film=None):
        n = w * h
        if film is None:
            we = np.zeros((n, 3), np.float32); acc = np.zeros(n, np.float64); smp = np.zeros(n, np.int64)
        else:
            we, acc, smp = film
        lib().orc_render(self._h, C.addressof(cam), w, h, C.c_uint64(seed), sample_begin, ns, integrator, we.ctypes.data,
                         acc.ctypes.data, smp.ctypes.data, nthreads)
        return we, acc, smp

    def counters(self):
        buf = (C.c_uint64 * len(COUNTER_NAMES))()
        lib().orc_counters_read(self._h, buf)
        return {k: int(buf[i]) for i, k in enumerate(COUNTER_NAMES)}

    def counters_reset(self):
        lib().orc_counters_reset(self._h)

    def bvh_info(self):
        a, b = C.c_int64(), C.c_int64()
        lib().orc_bvh_info(self._h, C.byref(a), C.byref(b))
        return {"nodes": a.value, "handles": b.value}


# unit hooks (orc_api.cpp orc_unit_eval; pathtracer_amd.Renderer.unit_eval is the device twin): kind -> (id, in width, out width)
UNIT_KINDS = {"fr_dielectric": (0, 3, 1), "fr_conductor": (1, 10, 3), "ggx_d": (2, 5, 1), "ggx_g1": (3, 5, 1), "ggx_pdf": (4, 8, 1),
              "ggx_sample_wh": (5, 7, 3), "hg_p": (6, 2, 1), "hg_sample": (7, 6, 4), "rough_to_alpha": (8, 1, 1), "env_sample": (9, 2, 3),
              "env_pdf": (10, 2, 1), "sss_sr": (11, 1, 3), "sss_pdf_sr": (12, 2, 1), "sss_sample_sr": (13, 2, 1)}


def unit_eval(kind, rows, scene=None, prim=-1):
    """rows: [n, in width] float32 -> [n, out width]; scene: an OracleScene for the env_* / sss_* kinds"""
    kid, iw, ow = UNIT_KINDS[kind]
    a = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, iw)
    out = np.zeros((a.shape[0], ow), np.float32)
    rc = lib().orc_unit_eval(scene._h if scene is not None else None, int(prim), kid, a.shape[0], a.ctypes.data, iw, out.ctypes.data, ow)
    if rc:
        raise ValueError("orc_unit_eval(%s) failed" % kind)
    return out


def bsdf_eval(scene, prim, wo, wi, uv=None):
    """f and pdf (non-specular lobes) of primitive `prim`'s material at a canonical interaction (n = +z, dpdu = +x)"""
    wo = np.ascontiguousarray(wo, dtype=np.float32).reshape(-1, 3)
    wi = np.ascontiguousarray(wi, dtype=np.float32).reshape(-1, 3)
    n = wo.shape[0]
    uv = np.full((n, 2), 0.5, np.float32) if uv is None else np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
    f, pdf = np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    lib().orc_bsdf_eval(scene._h, int(prim), n, wo.ctypes.data, wi.ctypes.data, uv.ctypes.data, f.ctypes.data, pdf.ctypes.data)
    return f, pdf


def bsdf_sample(scene, prim, wo, u2, uv=None):
    """BSDF::Sample_f(BSDF_ALL): wi, f, pdf, sampled type flags"""
    wo = np.ascontiguousarray(wo, dtype=np.float32).reshape(-1, 3)
    u2 = np.ascontiguousarray(u2, dtype=np.float32).reshape(-1, 2)
    n = wo.shape[0]
    uv = np.full((n, 2), 0.5, np.float32) if uv is None else np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
    wi, f, pdf, fl = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
    lib().orc_bsdf_sample(scene._h, int(prim), n, wo.ctypes.data, u2.ctypes.data, uv.ctypes.data, wi.ctypes.data, f.ctypes.data,
                          pdf.ctypes.data, fl.ctypes.data)
    return wi, f, pdf, fl


def image_of(we, acc):
    with np.errstate(divide="ignore", invalid="ignore"):
        img = we / acc[:, None].astype(np.float32)
    img[~np.isfinite(img)] = 0
    return img


def film_rgb8(we, acc, exposure=2.0, algorithm=1):
    """GetPixelRGB + ImageWrite's byte narrowing over a film (orc_film.cpp); uint8 [..., 3]."""
    we = np.ascontiguousarray(we, dtype=np.float32)
    acc = np.ascontiguousarray(acc, dtype=np.float64)
    out = np.empty(we.shape, np.uint8)
    rc = lib().orc_film_rgb8(acc.size, we.ctypes.data, acc.ctypes.data, C.c_float(exposure), int(algorithm), out.ctypes.data)
    if rc:
        raise ValueError("unknown tone mapping algorithm")
    return out


def envmap_tables(rgb, scale=(1.0, 1.0, 1.0)):
    """The reference's light-map bring-up (orc_envmap.cpp: texture.cpp:111-260, infinite.cpp:7-39) over rgb[h, w, 3]:
    {"dims": (width, height, nu, nv), "texels", "func", "cdf", "func_int", "marg_cdf": float32 arrays, "marg_int"}."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    sc3 = np.asarray(scale, dtype=np.float32)
    e = lib().orc_envmap_build(rgb.ctypes.data, w, h, sc3.ctypes.data)
    if not e:
        raise ValueError("orc_envmap_build refused the map")
    dims = (C.c_int32 * 4)()
    ptrs = [C.POINTER(C.c_float)() for _ in range(5)]
    mi = C.c_float()
    lib().orc_envmap_view(e, dims, *[C.byref(p) for p in ptrs], C.byref(mi))
    W, H, nu, nv = [int(v) for v in dims]
    sizes = (3 * W * H, nu * nv, (nu + 1) * nv, nv, nv + 1)
    out = {"dims": (W, H, nu, nv), "marg_int": np.float32(mi.value)}
    for k, p, n in zip(("texels", "func", "cdf", "func_int", "marg_cdf"), ptrs, sizes):
        out[k] = np.ctypeslib.as_array(p, (n,)).copy()
    lib().orc_envmap_free(e)
    return out
