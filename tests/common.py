"""Shared helpers of the parity tests: scene catalogue, oracle/HIP drivers, comparisons."""
import numpy as np

import pathtracer_amd as pa
from oracle import orc

# (family, p0, p1): small instances of every scene family the device path supports
SCENES = {
    "cornell": ("cornell", 0, 0),
    "cornell_glass": ("cornell", 0, 1),
    "cornell_fog": ("cornell", 0, 3),
    "knot_glass": ("knot", 24, 0),
    "knot_plastic": ("knot", 24, 1),
    "knot_specular": ("knot", 24, 2),
    "knot_matte": ("knot", 24, 3),
    "zoo": ("zoo", 0, 0),
    "zoo_sky_fog": ("zoo", 0, 3),
    "volcaustic": ("volcaustic", 0, 0),
    "fluid": ("fluid", 3000, 0),
    "origami": ("origami", 0, 0),
    "subsurface": ("subsurface", 0, 0),
    "subsurface_knot": ("subsurface", 8, 0),
    "cornell_disney": ("cornell", 0, 4),  # the boxes in the Disney material (PTR_MAT_DISNEY: the reference's five Disney lobes)
    "room": ("room", 6, 0),  # configs[2]: 40 rectangle / SDF-box primitives, 8 meshes, 64 emitters, image-textured walls
}
INTEGRATORS = {"direct": pa.LI_DIRECT, "path": pa.LI_PATH, "pathsampled": pa.LI_PATHSAMPLED, "volpath": pa.LI_VOLPATH}


def build(name, w, h):
    fam, p0, p1 = SCENES[name]
    return pa.BuiltScene(fam, w, h, p0, p1)


def oracle_of(scene):
    return orc.OracleScene(scene.flat, scene.rand_calls)


def random_rays(n, seed, lo=-3.0, hi=3.0):
    """Rays between random point pairs in a box (SoA planes [3, n])."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(lo, hi, size=(3, n)).astype(np.float32)
    b = rng.uniform(lo, hi, size=(3, n)).astype(np.float32)
    d = b - a
    nrm = np.sqrt((d.astype(np.float64) ** 2).sum(0)).astype(np.float32)
    nrm[nrm == 0] = 1
    d = (d / nrm).astype(np.float32)
    return a, d


def gpu_trace_closest(r, o, d, tmax):
    import torch

    n = o.shape[1]
    dev = torch.device("cuda:0")
    to = torch.from_numpy(np.ascontiguousarray(o)).to(dev)
    td = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
    tt = torch.from_numpy(np.ascontiguousarray(tmax)).to(dev)
    prim = torch.empty(n, dtype=torch.int32, device=dev)
    face = torch.empty(n, dtype=torch.int32, device=dev)
    t = torch.empty(n, dtype=torch.float32, device=dev)
    b0 = torch.empty(n, dtype=torch.float32, device=dev)
    b1 = torch.empty(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    r.trace_closest(n, to.data_ptr(), td.data_ptr(), tt.data_ptr(), prim.data_ptr(), face.data_ptr(), t.data_ptr(), b0.data_ptr(),
                    b1.data_ptr())
    r.sync()
    return {"prim": prim.cpu().numpy(), "face": face.cpu().numpy(), "t": t.cpu().numpy(), "b0": b0.cpu().numpy(),
            "b1": b1.cpu().numpy()}


def gpu_trace_occluded(r, o, d, tmax):
    import torch

    n = o.shape[1]
    dev = torch.device("cuda:0")
    to = torch.from_numpy(np.ascontiguousarray(o)).to(dev)
    td = torch.from_numpy(np.ascontiguousarray(d)).to(dev)
    tt = torch.from_numpy(np.ascontiguousarray(tmax)).to(dev)
    occ = torch.empty(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    r.trace_occluded(n, to.data_ptr(), td.data_ptr(), tt.data_ptr(), occ.data_ptr())
    r.sync()
    return occ.cpu().numpy()


def gpu_generate_rays(r, npix, sample):
    import torch

    dev = torch.device("cuda:0")
    o = torch.empty((3, npix), dtype=torch.float32, device=dev)
    d = torch.empty((3, npix), dtype=torch.float32, device=dev)
    w = torch.empty(npix, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    r.generate_rays(sample, o.data_ptr(), d.data_ptr(), w.data_ptr())
    r.sync()
    return o.cpu().numpy(), d.cpu().numpy(), w.cpu().numpy()


def image_stats(gpu_img, ref_img):
    """(relative error of the image mean, relative RMSE, fraction of pixels off by > 1e-3 relative)"""
    g = gpu_img.reshape(-1, 3).astype(np.float64)
    r = ref_img.reshape(-1, 3).astype(np.float64)
    mean_rel = abs(g.mean() - r.mean()) / max(r.mean(), 1e-12)
    rmse_rel = np.sqrt(((g - r) ** 2).mean()) / max(np.sqrt((r ** 2).mean()), 1e-12)
    denom = np.maximum(np.abs(r), 1e-3)
    off = (np.abs(g - r) / denom > 1e-3).any(axis=1).mean()
    return mean_rel, rmse_rel, off
