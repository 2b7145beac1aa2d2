"""What would sorting incoherent rays buy the tracer?  Bounce rays of the bench scene (origins on the first hits, random directions,
8 copies = 14.4 M rays) traced in pixel order, shuffled, and sorted by (origin cell, direction octant) keys of several widths.
usage: gpu_sortbench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import pathtracer_amd as pa

W, H = 1920, 1080
FMAX = np.finfo(np.float32).max
sc = pa.BuiltScene("knot", W, H, 660, 0)
r = pa.Renderer(0); r.upload(sc); r.set_camera(sc.camera); r.film_init(W, H, 1)
dev = torch.device("cuda:0")
n = W * H
o = torch.empty((3, n), dtype=torch.float32, device=dev); d = torch.empty_like(o); w = torch.empty(n, dtype=torch.float32, device=dev)
r.generate_rays(0, o.data_ptr(), d.data_ptr(), w.data_ptr()); r.sync()

def run(name, o, d, tmax, any_hit=False, reps=4):
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
    print("%-44s n %8d  %7.3f ms  %8.1f Mrays/s" % (name, n, ms, n / ms / 1e3), flush=True)
    return prim, t

tmax = torch.full((n,), FMAX, dtype=torch.float32, device=dev)
prim, t = run("primary (coherent)", o, d, tmax)
g = torch.Generator(device=dev); g.manual_seed(1)
idx = torch.nonzero(prim >= 0).squeeze(1)
p = o[:, idx] + d[:, idx] * t[idx]
p = p.repeat(1, 8)
v = torch.randn(p.shape, device=dev, generator=g); v = v / v.norm(dim=0, keepdim=True)
p2 = (p + 1e-3 * v).contiguous(); v = v.contiguous()
m = p2.shape[1]
tm = torch.full((m,), FMAX, dtype=torch.float32, device=dev)

def morton(q, bits):
    key = torch.zeros(q.shape[1], dtype=torch.int64, device=dev)
    for b in range(bits):
        for a in range(3):
            key |= ((q[a] >> b) & 1) << (3 * b + a)
    return key

def keys(bits, octant_first=False, dir_bits=1):
    lo = p2.min(dim=1, keepdim=True).values; hi = p2.max(dim=1, keepdim=True).values
    q = ((p2 - lo) / (hi - lo + 1e-9) * (1 << bits)).long().clamp(0, (1 << bits) - 1)
    mk = morton(q, bits)
    if dir_bits == 1:
        dk = (v[0] > 0).long() | ((v[1] > 0).long() << 1) | ((v[2] > 0).long() << 2)
        dn = 3
    else:
        qd = ((v * 0.5 + 0.5) * (1 << dir_bits)).long().clamp(0, (1 << dir_bits) - 1)
        dk = morton(qd, dir_bits); dn = 3 * dir_bits
    return (dk << (3 * bits)) | mk if octant_first else (mk << dn) | dk

for any_hit in (False, True):
    tag = "any " if any_hit else "closest "
    run(tag + "in pixel order", p2, v, tm, any_hit)
    perm = torch.randperm(m, device=dev, generator=g)
    run(tag + "shuffled", p2[:, perm].contiguous(), v[:, perm].contiguous(), tm, any_hit)
    for bits, of, db in ((4, False, 1), (6, False, 1), (8, False, 1), (10, False, 1), (6, True, 1), (6, False, 2), (6, True, 3)):
        perm = torch.argsort(keys(bits, of, db))
        run(tag + "sorted: origin %d bits/axis, dir %d bits/axis%s" % (bits, db, ", direction first" if of else ""), p2[:, perm].contiguous(), v[:, perm].contiguous(), tm, any_hit)
