"""First-contact GPU check: trace + image parity per scene / integrator, printed as a table."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import pathtracer_amd as pa
from oracle import orc
import common

W, H, SPP, SEED = 64, 48, 8, 7
names = sys.argv[1:] or list(common.SCENES)
for name in names:
    sc = common.build(name, W, H)
    o = common.oracle_of(sc)
    r = pa.Renderer(0)
    t0 = time.time()
    r.upload(sc)
    r.set_camera(sc.camera)
    r.film_init(W, H, SEED)
    print("==", name, "upload %.2fs" % (time.time() - t0), r.bvh_info(), flush=True)
    ro, rd, wt = common.gpu_generate_rays(r, W * H, 0)
    oo, od, ow = o.generate_rays(sc.camera, W, H, SEED, 0)
    print("   raygen: o exact", np.array_equal(ro, oo), "d exact", np.array_equal(rd, od), "w maxrel", np.max(np.abs(wt - ow) / np.maximum(ow, 1e-9)), flush=True)
    tmax = np.full(W * H, np.finfo(np.float32).max, np.float32)
    g = common.gpu_trace_closest(r, oo, od, tmax)
    c = o.trace_closest(oo, od, tmax)
    ok = c["order_dep"] == 0
    print("   closest(camera): prim mismatches", int(((g["prim"] != c["prim"]) & ok).sum()), "face", int(((g["face"] != c["face"]) & ok).sum()),
          "order_dep", int((~ok).sum()), "hits", int((c["prim"] >= 0).sum()),
          "t maxrel", float(np.max(np.where(ok & (c["prim"] >= 0), np.abs(g["t"] - c["t"]) / np.maximum(np.abs(c["t"]), 1e-9), 0))), flush=True)
    for integ_name, integ in common.INTEGRATORS.items():
        r.film_init(W, H, SEED)
        r.counters_reset()
        t0 = time.time()
        r.render(SPP, integ)
        r.sync()
        tg = time.time() - t0
        img = r.image()
        o.counters_reset()
        t0 = time.time()
        we, acc, smp = o.render(sc.camera, W, H, SEED, 0, SPP, integ, 8)
        tc = time.time() - t0
        ref = orc.image_of(we, acc).reshape(H, W, 3)
        m, rm, off = common.image_stats(img, ref)
        gc, oc = r.counters(), o.counters()
        cd = {k: (gc[k], oc[k]) for k in ("total_paths", "shadow_rays", "mis_rays", "zero_radiance", "medium_hits") if gc[k] != oc[k]}
        print("   %-11s mean_rel %.2e rmse_rel %.2e off %.4f nan %d  gpu %.2fs cpu %.2fs  ctr-diff %s" % (integ_name, m, rm, off, int(np.isnan(img).sum()), tg, tc, cd), flush=True)
    r.close()
