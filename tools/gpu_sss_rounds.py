"""How long the BSSRDF probe chains of configs[4] are: chains alive after each tracer round, per staged bounce of one pass.
usage (GPU box): PTR_SSS_ROUNDS=12 python3 tools/gpu_sss_rounds.py [spp]"""
import ctypes as C
import sys

import pathtracer_amd as pa
from pathtracer_amd import capi

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = pa.BuiltScene("subsurface", 1920, 1080, 387, 0)
r = pa.Renderer(0)
r.upload(sc)
r.set_camera(sc.camera)
r.film_init(1920, 1080, 1)
r.render(spp, pa.LI_VOLPATH)
r.sync()
lib = capi.hip()
lib.ptr_debug_sss_counts.argtypes = [capi.CTX, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
buf = (C.c_int * 1024)()
stride = C.c_int(0)
capi.check(lib.ptr_debug_sss_counts(r.ctx, 0, buf, 1024, C.byref(stride)))
st = stride.value
rounds = (st - 2) // 2
print("counters:", r.counters())
for k in range(12):
    row = buf[k * st:(k + 1) * st]
    if row[0] == 0:
        continue
    print("bounce %2d: deferred %9d  chains %9d  alive after round r:" % (k, row[0], row[1]), " ".join("%d" % v for v in row[2:2 + rounds]))
