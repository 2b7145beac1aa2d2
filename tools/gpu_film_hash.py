"""sha1 of the rendered film per (scene, integrator): compare two builds whose arithmetic is meant to be identical.
usage: PYTHONPATH=.:tests python3 tools/gpu_film_hash.py > hashes.txt"""
import hashlib, sys
import numpy as np
import pathtracer_amd as pa
sys.path.insert(0, "tests")
import common

W, H = 96, 64
for name in common.SCENES:
    sc = common.build(name, W, H)
    r = pa.Renderer(0)
    r.upload(sc)
    r.set_camera(sc.camera)
    for iname, integ in common.INTEGRATORS.items():
        r.film_init(W, H, 11)
        r.render(6, integ)
        we, acc, smp = r.film()
        print(name, iname, hashlib.sha1(np.ascontiguousarray(we).tobytes()).hexdigest()[:16], "%.6f" % float(we.mean()), flush=True)
    r.close()
    sc.close()
