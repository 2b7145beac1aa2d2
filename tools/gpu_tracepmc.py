"""One fixed tracer workload for counter passes: the shuffled bounce rays of tools/gpu_tracebench.py, x8
(14.4 M rays), traced REPS times closest-hit and REPS times any-hit.  Prints ms per launch.
Run under `rocprofv3 --pmc ...` (tools/gpu_tracepmc.sh); PTR_TRACER=lane|quad picks the tracer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pathtracer_amd as pa

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W, H = 1920, 1080
FMAX = float(np.finfo(np.float32).max)
sc = pa.BuiltScene("knot", W, H, 660, 0)
os.environ.setdefault("PTR_PASS_SLOTS", "2073600")  # a one-sample pass buffer is all this needs
r = pa.Renderer(0); r.upload(sc); r.set_camera(sc.camera)
r.film_init(W, H, 1)
dev = torch.device("cuda:0")
n = W * H
o = torch.empty((3, n), dtype=torch.float32, device=dev); d = torch.empty_like(o); w = torch.empty(n, dtype=torch.float32, device=dev)
r.generate_rays(0, o.data_ptr(), d.data_ptr(), w.data_ptr()); r.sync()
tmax = torch.full((n,), FMAX, dtype=torch.float32, device=dev)
prim = torch.empty(n, dtype=torch.int32, device=dev); face = torch.empty_like(prim)
t = torch.empty(n, dtype=torch.float32, device=dev); b0 = torch.empty_like(t); b1 = torch.empty_like(t)
r.trace_closest(n, o.data_ptr(), d.data_ptr(), tmax.data_ptr(), prim.data_ptr(), face.data_ptr(), t.data_ptr(), b0.data_ptr(), b1.data_ptr()); r.sync()
g = torch.Generator(device=dev); g.manual_seed(1)
idx = torch.nonzero(prim == prim.max()).squeeze(1)
p = o[:, idx] + d[:, idx] * t[idx]
v = torch.randn(p.shape, device=dev, generator=g); v = v / v.norm(dim=0, keepdim=True)
p2 = (p + 1e-3 * v)
perm = torch.randperm(p2.shape[1], device=dev, generator=g)
P = p2[:, perm].repeat(1, 8).contiguous(); V = v[:, perm].repeat(1, 8).contiguous()
m = P.shape[1]
TM = torch.full((m,), FMAX, dtype=torch.float32, device=dev)
prim = torch.empty(m, dtype=torch.int32, device=dev); face = torch.empty_like(prim)
t = torch.empty(m, dtype=torch.float32, device=dev); b0 = torch.empty_like(t); b1 = torch.empty_like(t)
torch.cuda.synchronize()
for any_hit in (False, True):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        if any_hit: r.trace_occluded(m, P.data_ptr(), V.data_ptr(), TM.data_ptr(), prim.data_ptr())
        else: r.trace_closest(m, P.data_ptr(), V.data_ptr(), TM.data_ptr(), prim.data_ptr(), face.data_ptr(), t.data_ptr(), b0.data_ptr(), b1.data_ptr())
    e1.record(); torch.cuda.synchronize()
    print("%s %s n %d  %.3f ms per launch" % (os.environ.get("PTR_TRACER", "default"), "any" if any_hit else "closest", m, e0.elapsed_time(e1) / REPS), flush=True)
r.close()
