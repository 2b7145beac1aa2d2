"""repeatability at full size: the same render several times in one process (six lanes, passes finishing in any order) must give
the same film bytes: PYTHONPATH=. python3 tools/gpu_soak.py [config] [spp] [repeats]"""
import hashlib, sys, time
import numpy as np
import pathtracer_amd as pa
sys.path.insert(0, ".")
from bench import CONFIGS

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fam, p0, p1, W, H, spp, label = CONFIGS[cfg]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2 * spp
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sc = pa.BuiltScene(fam, W, H, p0, p1)
hashes = []
with pa.Renderer(0) as r:
    r.upload(sc)
    r.set_camera(sc.camera)
    for i in range(reps):
        r.film_init(W, H, 3)
        t = time.time()
        r.render(spp, 3)
        r.sync()
        dt = time.time() - t
        we, acc, smp = r.film()
        hashes.append(hashlib.sha1(np.ascontiguousarray(we).tobytes()).hexdigest()[:16])
        print("config %d run %d: %.2f s, film %s, finite %s" % (cfg, i, dt, hashes[-1], bool(np.isfinite(we).all())), flush=True)
print("REPEATABLE" if len(set(hashes)) == 1 else "DIFFERENT FILMS")
