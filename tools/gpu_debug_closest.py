import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import pathtracer_amd as pa
import common
W, H, SEED = 64, 48, 7
FMAX = np.finfo(np.float32).max
for name in sys.argv[1:] or ["cornell"]:
    sc = common.build(name, W, H); o = common.oracle_of(sc)
    r = pa.Renderer(0); r.upload(sc); r.set_camera(sc.camera); r.film_init(W, H, SEED)
    for sample in range(4):
        oo, od, _ = o.generate_rays(sc.camera, W, H, SEED, sample)
        tm = np.full(W * H, FMAX, np.float32)
        g = common.gpu_trace_closest(r, oo, od, tm); c = o.trace_closest(oo, od, tm)
        bad = np.nonzero((g["prim"] != c["prim"]) | (g["face"] != c["face"]) | (g["t"] != c["t"]))[0]
        print(name, "sample", sample, "mismatch", bad.size, "order_dep among them", int(c["order_dep"][bad].sum()))
        for i in bad[:10]:
            print("   ray", i, "o", oo[:, i], "d", od[:, i], "gpu", g["prim"][i], g["face"][i], repr(g["t"][i]), "orc", c["prim"][i], c["face"][i], repr(c["t"][i]), "od", c["order_dep"][i])
