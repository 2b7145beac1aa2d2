import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import pathtracer_amd as pa
import common
W, H = 960, 540
sc = pa.BuiltScene("knot", W, H, 660, 0)
r = pa.Renderer(0); r.upload(sc); r.set_camera(sc.camera); r.film_init(W, H, 1)
for integ in (1, 2, 3):
    r.film_init(W, H, 1); r.counters_reset()
    t0 = time.time(); r.render(2, integ); r.sync(); dt = time.time() - t0
    we, acc, smp = r.film()
    print("integ", integ, "time %.3f" % dt, "nan pixels", int(np.isnan(we).any(axis=2).sum()), "inf", int(np.isinf(we).any(axis=2).sum()), r.counters())
# time a NaN ray and an ordinary ray through the stand-alone entry point
for name, d in (("nan", [np.nan, 0, 1]), ("ordinary", [0.1, -0.3, 1.0]), ("inf", [np.inf, 0, 1]), ("zero", [0, 0, 0])):
    o = np.array([[0.], [1.], [-5.]], np.float32); dd = np.array(d, np.float32).reshape(3, 1)
    tm = np.full(1, np.finfo(np.float32).max, np.float32)
    common.gpu_trace_closest(r, o, dd, tm)
    t0 = time.time(); g = common.gpu_trace_closest(r, o, dd, tm); dt = time.time() - t0
    print(name, "ray: %.3f ms" % (dt * 1e3), g["prim"], g["face"], g["t"])
