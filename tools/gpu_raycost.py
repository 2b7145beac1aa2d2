"""Distribution of traversal work per ray (child boxes / leaf items tested) on the bench scene."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import pathtracer_amd as pa
import common
W, H = 960, 540
FMAX = np.finfo(np.float32).max
sc = pa.BuiltScene("knot", W, H, int(sys.argv[1]) if len(sys.argv) > 1 else 660, 0)
r = pa.Renderer(0); r.upload(sc); r.set_camera(sc.camera); r.film_init(W, H, 1)
o, d, _ = common.gpu_generate_rays(r, W * H, 0)
tm = np.full(W * H, FMAX, np.float32)
g = common.gpu_trace_closest(r, o, d, tm)
hit = g["prim"] == np.max(g["prim"])  # the mesh
print("primary: hits on mesh", int(hit.sum()))
rng = np.random.default_rng(0)
def stats(name, o, d):
    tm = np.full(o.shape[1], FMAX, np.float32)
    r.profile_enable(False, True)
    c = common.gpu_trace_closest(r, o, d, tm)
    r.profile_enable(False, False)
    t0 = time.time(); common.gpu_trace_closest(r, o, d, tm); dt = time.time() - t0
    nodes, items = c["b0"], c["b1"]
    q = [50, 90, 99, 99.9, 99.99, 100]
    print(name, "n", o.shape[1], "mean nodes %.1f items %.1f" % (nodes.mean(), items.mean()),
          "nodes pct", dict(zip(q, np.percentile(nodes, q).astype(int))), "items pct", dict(zip(q, np.percentile(items, q).astype(int))))
    # per-wave max vs mean (64 consecutive rays)
    n64 = (nodes.size // 64) * 64
    w = (nodes[:n64] + 3 * items[:n64]).reshape(-1, 64)
    print("   wave max/mean cost ratio %.2f  (wall %.1f ms incl. copies)" % (w.max(1).mean() / w.mean(), dt * 1e3))
    return nodes, items
stats("primary", o, d)
# bounce rays: from mesh hit points, uniformly random directions; shuffled (incoherent)
idx = np.nonzero(hit)[0]
p = o[:, idx] + d[:, idx] * g["t"][idx]
v = rng.normal(size=p.shape).astype(np.float32); v /= np.linalg.norm(v, axis=0)
p2 = (p + 1e-3 * v).astype(np.float32)
perm = rng.permutation(p2.shape[1])
nodes, items = stats("bounce(random dir, shuffled)", np.ascontiguousarray(p2[:, perm]), np.ascontiguousarray(v[:, perm]))
worst = np.argsort(nodes)[-5:]
print("worst rays nodes", nodes[worst], "items", items[worst])
