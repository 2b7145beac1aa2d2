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
        L.orc_film_rgb8.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
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
