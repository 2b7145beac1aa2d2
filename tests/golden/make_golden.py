"""Generates tests/golden/*.npz from the oracle (oracle/liborc.so) - run from the repo root:
    python tests/golden/make_golden.py
The reference holds no test vectors for this path and cannot be compiled in this environment
(CUDA / cuRAND headers), so these are regression pins of the CPU restatement: inputs (primary
rays) and expected outputs (hit records, 2-spp film per integrator)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402

W, H, SEED = 32, 24, 1234
FMAX = np.finfo(np.float32).max
for name in (sys.argv[1:] or ["cornell", "knot_glass", "zoo_sky_fog", "fluid", "volcaustic", "room", "cornell_disney"]):
    sc = common.build(name, W, H)
    o = common.oracle_of(sc)
    oo, od, ow = o.generate_rays(sc.camera, W, H, SEED, 0)
    res = o.trace_closest(oo, od, np.full(W * H, FMAX, np.float32))
    out = dict(width=W, height=H, seed=SEED, ray_o=oo, ray_d=od, weight=ow, **res)
    for integ in range(4):
        we, acc, _ = o.render(sc.camera, W, H, SEED, 0, 2, integ, 4)
        out["we_%d" % integ] = we
        out["acc"] = acc
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)
    print(name, "hits", int((res["prim"] >= 0).sum()), "order_dep", int(res["order_dep"].sum()))
