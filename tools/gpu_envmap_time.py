"""host against device build of the light-map tables (InsertLightMap pipeline), wall time per map size"""
import ctypes as C, sys, time
import numpy as np
from pathtracer_amd import capi

for w, h in ((1024, 512), (2048, 1024), (4096, 2048), (3000, 1500)):
    img = np.random.default_rng(1).uniform(0, 1, (h, w, 3)).astype(np.float32)
    sc3 = (C.c_float * 3)(1, 1, 1)
    p = img.ctypes.data_as(C.POINTER(C.c_float))
    t = {}
    for mode in (1, 0, 1):
        t0 = time.perf_counter()
        hnd = capi.host().pth_lightmap_scene(p, w, h, sc3, mode)
        t[mode] = time.perf_counter() - t0
        capi.host().pth_scene_free(hnd)
    out = C.c_void_p()
    capi.check(capi.hip().ptr_envmap_build(-1, p, w, h, sc3, C.byref(out)))
    v, ms = capi.PtrEnvMap(), C.c_double()
    capi.hip().ptr_envmap_view(out, C.byref(v), C.byref(ms))
    capi.hip().ptr_envmap_free(out)
    print("%4d x %4d: host %7.1f ms   device %6.1f ms wall (kernels %.3f ms, the rest is copies in/out + flatten)" % (w, h, 1e3 * t[0], 1e3 * t[1], ms.value), flush=True)
