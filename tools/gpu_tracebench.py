"""Tracer microbenchmark: ptr_trace_closest / ptr_trace_occluded on fixed ray sets of the bench
scene, timed with events on the launch stream.  usage: gpu_tracebench.py [segments] [WxH]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import pathtracer_amd as pa
import common

seg = int(sys.argv[1]) if len(sys.argv) > 1 else 660
W, H = 1920, 1080
FMAX = np.finfo(np.float32).max
sc = pa.BuiltScene("knot", W, H, seg, 0)
r = pa.Renderer(0); r.upload(sc); r.set_camera(sc.camera); r.film_init(W, H, 1)
dev = torch.device("cuda:0")
n = W * H
o = torch.empty((3, n), dtype=torch.float32, device=dev); d = torch.empty_like(o); w = torch.empty(n, dtype=torch.float32, device=dev)
r.generate_rays(0, o.data_ptr(), d.data_ptr(), w.data_ptr()); r.sync()

def run(name, o, d, tmax, any_hit=False, reps=5):
    n = o.shape[1]
    prim = torch.empty(n, dtype=torch.int32, device=dev); face = torch.empty_like(prim)
    t = torch.empty(n, dtype=torch.float32, device=dev); b0 = torch.empty_like(t); b1 = torch.empty_like(t)
    def launch():
        if any_hit: r.trace_occluded(n, o.data_ptr(), d.data_ptr(), tmax.data_ptr(), prim.data_ptr())
        else: r.trace_closest(n, o.data_ptr(), d.data_ptr(), tmax.data_ptr(), prim.data_ptr(), face.data_ptr(), t.data_ptr(), b0.data_ptr(), b1.data_ptr())
    launch(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): launch()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    r.profile_enable(False, True); r.counters_reset()
    if not any_hit:
        launch(); r.sync()
        nodes, items = b0.mean().item(), b1.mean().item()
    else: nodes = items = float("nan")
    r.profile_enable(False, False)
    launch(); r.sync()
    print("%-34s n %8d  %7.3f ms  %8.1f Mrays/s  nodes/ray %6.1f items/ray %5.1f" % (name, n, ms, n / ms / 1e3, nodes, items), flush=True)
    return prim, t

tmax = torch.full((n,), FMAX, dtype=torch.float32, device=dev)
prim, t = run("primary (coherent)", o, d, tmax)
# bounce rays from the hit points of the mesh, random directions, shuffled (incoherent)
g = torch.Generator(device=dev); g.manual_seed(1)
mesh = prim == prim.max()
idx = torch.nonzero(mesh).squeeze(1)
p = o[:, idx] + d[:, idx] * t[idx]
v = torch.randn(p.shape, device=dev, generator=g); v = v / v.norm(dim=0, keepdim=True)
p2 = (p + 1e-3 * v).contiguous(); v = v.contiguous()
perm = torch.randperm(p2.shape[1], device=dev, generator=g)
tm2 = torch.full((p2.shape[1],), FMAX, dtype=torch.float32, device=dev)
run("bounce (from knot, in order)", p2, v, tm2)
run("bounce (from knot, shuffled)", p2[:, perm].contiguous(), v[:, perm].contiguous(), tm2)
run("any-hit (from knot, shuffled)", p2[:, perm].contiguous(), v[:, perm].contiguous(), tm2, any_hit=True)
small = perm[:4096]
run("bounce, 4096 rays", p2[:, small].contiguous(), v[:, small].contiguous(), tm2[:4096].contiguous())

# same rays against the mesh alone (top level = one leaf): what the top-level walk + ground sphere cost
fs = sc.flat.contents
me = fs.meshes[0]
verts = np.ctypeslib.as_array(me.p, shape=(me.n_vertices * 3,)).reshape(-1, 3).copy()
idxs = np.ctypeslib.as_array(me.indices, shape=(me.n_triangles * 9,)).reshape(-1, 3, 3)[:, :, 0].copy()
m4 = np.array(list(me.to_world.m), np.float64).reshape(4, 4)
b = pa.FlatSceneBuilder()
b.mesh(verts, idxs, pa.material("glass", index=1.5, uroughness=0.05, vroughness=0.05), to_world=m4)
sc2 = b.finish()
r2 = pa.Renderer(0); r2.upload(sc2); r2.set_camera(sc.camera); r2.film_init(W, H, 1)
r = r2
print("-- mesh only")
run("primary (coherent)", o, d, tmax)
run("bounce (from knot, shuffled)", p2[:, perm].contiguous(), v[:, perm].contiguous(), tm2)

# tail model: time(k x rays) = k * T_bulk + T_tail ?
r = pa.Renderer(0); r.upload(sc); r.set_camera(sc.camera); r.film_init(W, H, 1)
print("-- scaling with batch size (full scene)")
ob, db = p2[:, perm].contiguous(), v[:, perm].contiguous()
for k in (1, 2, 4, 8):
    run("bounce shuffled x%d" % k, ob.repeat(1, k).contiguous(), db.repeat(1, k).contiguous(), tm2.repeat(k).contiguous(), reps=3)
for k in (2, 4):
    run("primary x%d" % k, o.repeat(1, k).contiguous(), d.repeat(1, k).contiguous(), tmax.repeat(k).contiguous(), reps=3)
