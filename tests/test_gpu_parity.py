"""GPU parity: the HIP path, called through the C-ABI, against the CPU oracle on the same seeded
inputs.  Tolerances (BASELINE.json north_star): closest-hit primitive / face ids bit-exact on every
ray whose reference answer does not depend on the reference's own BVH order (exact-t ties, flagged
by the oracle); t and barycentrics within 1e-5 relative; images at equal spp within the stated
mean-relative-error / relative-RMSE bounds (device libm differs from glibc by ulps, which flips a
discrete sampling decision on a small fraction of paths)."""
import json
import os

import numpy as np
import pytest

import pathtracer_amd as pa
from oracle import orc

import common

pytestmark = pytest.mark.gpu
W, H, SEED = 64, 48, 7
FMAX = np.finfo(np.float32).max


@pytest.fixture(scope="module")
def scenes(cuda):
    cache = {}

    def get(name):
        if name not in cache:
            sc = common.build(name, W, H)
            r = pa.Renderer(0)
            r.upload(sc)
            r.set_camera(sc.camera)
            r.film_init(W, H, SEED)
            cache[name] = (sc, common.oracle_of(sc), r)
        return cache[name]

    yield get
    for sc, o, r in cache.values():
        r.close()
        o.close()


def check_closest(g, c, ray_scale=1.0):
    ok = c["order_dep"] == 0
    assert ok.mean() > 0.95
    assert np.array_equal(g["prim"][ok], c["prim"][ok])
    assert np.array_equal(g["face"][ok], c["face"][ok])
    hit = ok & (c["prim"] >= 0)
    rel = np.abs(g["t"][hit] - c["t"][hit]) / np.maximum(np.abs(c["t"][hit]), 1e-12)
    assert rel.size == 0 or rel.max() <= 1e-5
    for k in ("b0", "b1"):
        assert np.all(np.abs(g[k][hit] - c[k][hit]) <= 1e-5)
    miss = ok & (c["prim"] < 0)
    assert np.array_equal(g["t"][miss], c["t"][miss])  # tmax passes through untouched


@pytest.mark.parametrize("name", list(common.SCENES))
def test_generate_rays(scenes, name):
    sc, o, r = scenes(name)
    for sample in (0, 5):
        go, gd, gw = common.gpu_generate_rays(r, W * H, sample)
        oo, od, ow = o.generate_rays(sc.camera, W, H, SEED, sample)
        if sc.camera.lens_radius == 0:
            assert np.array_equal(go, oo) and np.array_equal(gd, od)
        else:  # thin lens: cosf/sinf of the device libm enter the origin
            assert np.allclose(go, oo, rtol=1e-5, atol=1e-6) and np.allclose(gd, od, rtol=1e-5, atol=1e-6)
        assert np.allclose(gw, ow, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", list(common.SCENES))
def test_primary_closest_hit(scenes, name):
    sc, o, r = scenes(name)
    oo, od, _ = o.generate_rays(sc.camera, W, H, SEED, 1)
    tmax = np.full(W * H, FMAX, np.float32)
    check_closest(common.gpu_trace_closest(r, oo, od, tmax), o.trace_closest(oo, od, tmax))


@pytest.mark.parametrize("name", ["cornell", "knot_glass", "zoo", "fluid"])
def test_incoherent_closest_hit(scenes, name):
    sc, o, r = scenes(name)
    n = 20000
    ro, rd = common.random_rays(n, 3, -4.0, 4.0)
    tmax = np.full(n, FMAX, np.float32)
    tmax[::7] = 2.5  # bounded segments as well
    check_closest(common.gpu_trace_closest(r, ro, rd, tmax), o.trace_closest(ro, rd, tmax))


@pytest.mark.parametrize("name", ["cornell", "knot_glass", "zoo", "fluid"])
def test_occluded(scenes, name):
    sc, o, r = scenes(name)
    n = 20000
    ro, rd = common.random_rays(n, 5, -4.0, 4.0)
    tmax = np.full(n, 3.0, np.float32)
    assert np.array_equal(common.gpu_trace_occluded(r, ro, rd, tmax), o.trace_occluded(ro, rd, tmax))


def test_empty_and_single_ray(scenes):
    sc, o, r = scenes("cornell")
    r.trace_closest(0, None, None, None, None, None, None)  # n = 0 is a no-op
    ro, rd = common.random_rays(1, 9)
    tmax = np.full(1, FMAX, np.float32)
    check_closest(common.gpu_trace_closest(r, ro, rd, tmax), o.trace_closest(ro, rd, tmax))


# image parity: (mean relative error, relative RMSE, fraction of pixels off by > 1e-3 relative).
# Bounds per case = 10 x the residual measured on the MI355X (tests/image_bounds.json, written by this test when
# PT_RECORD_IMAGE_STATS names a file, then scaled by tests/make_image_bounds.py), never looser than the ceiling below.
# The residual comes from device libm (sinf / cosf / expf / logf / powf / atan2f / acosf differ from glibc by ulps), which
# flips a discrete sampling decision on a fraction of a per cent of the paths.
IMAGE_CEILING = (5e-3, 5e-2, 0.02)
IMAGE_FLOOR = (1e-5, 2e-4, 4e-3)  # one flipped sampling decision in a 32 x 24 frame moves a case by about this much
_bounds_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image_bounds.json")
IMAGE_BOUNDS = json.load(open(_bounds_path)) if os.path.exists(_bounds_path) else {}
IMAGE_CASES = [(s, i) for s in common.SCENES for i in common.INTEGRATORS]
@pytest.mark.parametrize("name,integ", IMAGE_CASES)
def test_image(scenes, name, integ):
    sc, o, r = scenes(name)
    spp = 8
    r.film_init(W, H, SEED)
    r.counters_reset()
    r.render(spp, common.INTEGRATORS[integ])
    r.sync()
    img = r.image()
    _, _, smp = r.film()
    o.counters_reset()
    we, acc, osmp = o.render(sc.camera, W, H, SEED, 0, spp, common.INTEGRATORS[integ], 8)
    ref = orc.image_of(we, acc).reshape(H, W, 3)
    assert np.isfinite(img).all()
    assert np.array_equal(smp.reshape(-1), osmp)
    mean_rel, rmse_rel, off = common.image_stats(img, ref)
    rec = os.environ.get("PT_RECORD_IMAGE_STATS")
    if rec:
        with open(rec, "a") as f:
            f.write(json.dumps({"case": "%s/%s" % (name, integ), "mean_rel": mean_rel, "rmse_rel": rmse_rel, "off": off}) + "\n")
    bound = IMAGE_BOUNDS.get("%s/%s" % (name, integ), IMAGE_CEILING)
    bound = [min(max(b, f), c) for b, f, c in zip(bound, IMAGE_FLOOR, IMAGE_CEILING)]
    assert mean_rel <= bound[0], (mean_rel, rmse_rel, off, bound)
    assert rmse_rel <= bound[1], (mean_rel, rmse_rel, off, bound)
    assert off <= bound[2], (mean_rel, rmse_rel, off, bound)
    gc, oc = r.counters(), o.counters()
    assert gc["camera_rays"] == oc["camera_rays"] == W * H * spp
    for k in ("total_paths", "shadow_rays", "mis_rays", "surface_hits", "medium_hits", "misses"):
        assert abs(gc[k] - oc[k]) <= 0.005 * max(oc[k], 1) + 2, (k, gc[k], oc[k])


def test_render_in_batches_is_identical(scenes):
    """ns samples in one call == the same samples over several calls (per-sample RNG streams)."""
    sc, o, r = scenes("cornell_glass")
    r.film_init(W, H, SEED)
    r.render(6, pa.LI_PATHSAMPLED)
    a = r.film()[0].copy()
    r.film_init(W, H, SEED)
    for ns in (1, 2, 3):
        r.render(ns, pa.LI_PATHSAMPLED)
    b = r.film()[0]
    assert np.array_equal(a, b)


def test_sample_shards_partition_the_film(scenes):
    """shard (0,2) + shard (1,2) accumulate exactly the samples of the unsharded render."""
    sc, o, r = scenes("knot_plastic")
    r.film_init(W, H, SEED)
    r.shard(0, 1)
    r.render(4, pa.LI_PATH)
    full_we, full_acc, full_n = [x.copy() for x in r.film()]
    parts = []
    for off in (0, 1):
        r.film_init(W, H, SEED)
        r.shard(off, 2)
        r.render(4, pa.LI_PATH)
        parts.append([x.copy() for x in r.film()])
    r.shard(0, 1)
    assert np.array_equal(parts[0][2] + parts[1][2], full_n)
    assert np.allclose(parts[0][1] + parts[1][1], full_acc, rtol=1e-12)
    assert np.allclose(parts[0][0] + parts[1][0], full_we, rtol=1e-5, atol=1e-6)


def test_python_built_scene(cuda):
    """A scene assembled through FlatSceneBuilder: mesh + sphere + rect light, every integrator."""
    b = pa.FlatSceneBuilder()
    v = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1], [0, 1.2, 0]], np.float32)
    t = np.array([[0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4], [0, 2, 1], [0, 3, 2]], np.int32)
    from pathtracer_amd import scene as S
    b.mesh(v, t, pa.material("plastic", kd=(0.6, 0.2, 0.2)), to_world=S.translate(0, -0.5, 0))
    b.sphere(0.4, pa.material("glass", index=1.5), to_world=S.translate(1.2, 0.0, 0.3))
    b.rectangle(8, 8, pa.material("matte", kd=(0.7, 0.7, 0.7)), to_world=S.translate(0, -0.5, 0) @ S.rotate_x(-90))
    b.rectangle(1.5, 1.5, pa.material("emissive", L=(9, 9, 9)), to_world=S.translate(0, 3, 0) @ S.rotate_x(90))
    sc = b.finish()
    cam = pa.look_at_camera((0, 1.5, 5), (0, 0.2, 0), (0, 1, 0), 40, W / H)
    o = orc.OracleScene(sc.flat, 0)
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(cam)
        oo, od, _ = o.generate_rays(cam, W, H, SEED, 0)
        tmax = np.full(W * H, FMAX, np.float32)
        check_closest(common.gpu_trace_closest(r, oo, od, tmax), o.trace_closest(oo, od, tmax))
        for integ in common.INTEGRATORS.values():
            r.film_init(W, H, SEED)
            r.render(4, integ)
            img = r.image()
            we, acc, _ = o.render(cam, W, H, SEED, 0, 4, integ, 8)
            ref = orc.image_of(we, acc).reshape(H, W, 3)
            mean_rel, rmse_rel, off = common.image_stats(img, ref)
            assert mean_rel < 5e-3 and rmse_rel < 5e-2 and off < 0.02, (integ, mean_rel, rmse_rel, off)


def test_errors(cuda):
    with pa.Renderer(0) as r:
        with pytest.raises(pa.PtrError) as e:
            r.render(1)
        assert e.value.code == pa.capi.PTR_ERR_STATE
        with pytest.raises(pa.PtrError):
            r.film_init(0, 10)
        with pytest.raises(pa.PtrError):
            r.shard(2, 2)


def test_film_tensor_views_for_the_all_reduce(scenes):
    """pathtracer_amd.dist.film_tensors: torch views of the accumulators inside the library (what
    the RCCL all-reduce of bench.py operates on) alias the film the C-ABI reads back"""
    import torch
    from pathtracer_amd import dist as pdist

    sc, o, r = scenes("cornell")
    r.film_init(W, H, SEED)
    r.render(2, pa.LI_PATH)
    r.sync()
    we_t, acc_t, smp_t = pdist.film_tensors(r, torch.device("cuda:0"))
    we, acc, smp = r.film()
    assert np.array_equal(we_t.cpu().numpy(), we.reshape(-1))
    assert np.array_equal(acc_t.cpu().numpy(), acc.reshape(-1))
    assert np.array_equal(smp_t.cpu().numpy(), smp.reshape(-1))
    we_t.mul_(2.0)  # in place on the device: visible through ptr_film_read
    torch.cuda.synchronize()
    assert np.array_equal(r.film()[0], we * 2)


def test_thin_lens_camera(cuda):
    """Camera::Config with an aperture (camera.cpp:145-197, SpawnRay 219-229): lens samples move the ray origin; no
    shipped scene family uses it, so the rays and an image are compared on a scene assembled here."""
    from pathtracer_amd import scene as S
    b = pa.FlatSceneBuilder()
    b.rectangle(12, 12, pa.material("matte", kd=(0.6, 0.6, 0.6)), to_world=S.rotate_x(-90))
    b.sphere(0.7, pa.material("plastic", kd=(0.7, 0.2, 0.2)), to_world=S.translate(-0.8, 0.7, 0.5))
    b.sphere(0.5, pa.material("glass", index=1.5), to_world=S.translate(0.9, 0.5, -1.0))
    b.rectangle(2, 2, pa.material("emissive", L=(12, 12, 12)), to_world=S.translate(0, 4, 0) @ S.rotate_x(90))
    sc = b.finish()
    cam = pa.look_at_camera((0, 1.8, -5), (0, 0.5, 0), (0, 1, 0), 35, W / H, aperture=0.4)
    assert cam.lens_radius > 0
    o = orc.OracleScene(sc.flat, 0)
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(cam)
        r.film_init(W, H, SEED)
        for sample in (0, 3):
            go, gd, gw = common.gpu_generate_rays(r, W * H, sample)
            oo, od, ow = o.generate_rays(cam, W, H, SEED, sample)
            assert np.ptp(oo, axis=1).max() > 0.05  # origins really are spread over the lens
            # device cosf / sinf of the lens sample enter the origin: ulps, not bit-exact
            assert np.allclose(go, oo, rtol=1e-5, atol=1e-6) and np.allclose(gd, od, rtol=1e-5, atol=1e-6)
            assert np.allclose(gw, ow, rtol=1e-5, atol=1e-7)
        r.render(8, pa.LI_PATHSAMPLED)
        img = r.image()
        we, acc, _ = o.render(cam, W, H, SEED, 0, 8, pa.LI_PATHSAMPLED, 8)
        ref = orc.image_of(we, acc).reshape(H, W, 3)
        mean_rel, rmse_rel, off = common.image_stats(img, ref)
        assert mean_rel < 5e-3 and rmse_rel < 5e-2 and off < 0.02, (mean_rel, rmse_rel, off)
    o.close()


def test_caller_stream(scenes):
    """ptr_set_stream: the stand-alone trace runs on the caller's stream, and work the caller queues there after
    ptr_render sees the finished film without a host sync in between (the lanes' streams are joined into it)."""
    import torch
    from pathtracer_amd import dist as pdist
    sc, o, r = scenes("knot_plastic")
    dev = torch.device("cuda:0")
    oo, od, _ = o.generate_rays(sc.camera, W, H, SEED, 0)
    tmax = np.full(W * H, FMAX, np.float32)
    want = common.gpu_trace_closest(r, oo, od, tmax)  # on the context's default stream
    s = torch.cuda.Stream(device=dev)
    r.set_stream(s.cuda_stream)
    try:
        with torch.cuda.stream(s):
            n = W * H
            to = torch.from_numpy(oo).to(dev, non_blocking=True)
            td = torch.from_numpy(od).to(dev, non_blocking=True)
            tt = torch.from_numpy(tmax).to(dev, non_blocking=True)
            prim = torch.full((n,), -7, dtype=torch.int32, device=dev)
            face = torch.empty(n, dtype=torch.int32, device=dev)
            t = torch.empty(n, dtype=torch.float32, device=dev)
            r.trace_closest(n, to.data_ptr(), td.data_ptr(), tt.data_ptr(), prim.data_ptr(), face.data_ptr(), t.data_ptr())
            got_prim = prim.clone()  # ordered after the trace by the stream alone
        s.synchronize()
        assert np.array_equal(got_prim.cpu().numpy(), want["prim"])
        assert np.array_equal(t.cpu().numpy(), want["t"])

        r.film_init(W, H, SEED)
        we_t, acc_t, smp_t = pdist.film_tensors(r, dev)
        with torch.cuda.stream(s):
            r.render(6, pa.LI_PATH)
            early = we_t.double().sum()  # queued on the caller's stream right behind the render, no host sync
        s.synchronize()
        r.sync()
        assert float(early) == float(we_t.double().sum()) and float(early) > 0
        assert int(smp_t.min()) == 6

        # the other direction: what the caller queued on its stream BEFORE ptr_render (here a slow kernel chain that ends by
        # writing the accumulators, as an all-reduce or a film restore would) is in the film the passes add to
        r.film_init(W, H, SEED)
        r.render(6, pa.LI_PATH)
        r.sync()
        base_we, base_acc = we_t.clone(), acc_t.clone()
        r.film_init(W, H, SEED)
        big = torch.randn(4096, 4096, device=dev)
        with torch.cuda.stream(s):
            for _ in range(6):
                big = big @ big * 1e-3  # keeps the stream busy for a while
            we_t.fill_(2.0)
            acc_t.fill_(3.0)
            r.render(6, pa.LI_PATH)  # no host sync between the fills and the render
        s.synchronize()
        r.sync()
        assert torch.allclose(we_t, base_we + 2.0, rtol=1e-6, atol=1e-6)
        assert torch.allclose(acc_t, base_acc + 3.0, rtol=1e-12, atol=1e-9)
    finally:
        r.set_stream(0)


def test_film_init_shrinks_and_recovers(scenes):
    """ptr_film_init sizes its pass buffers to what the device has: a target far beyond the memory (PTR_PASS_SLOTS) ends
    in a smaller pass, not in an error, the render is the same image, and a film that cannot be allocated at all leaves
    nothing behind (a normal film can be made right after)."""
    import os
    import torch
    sc, o, r = scenes("cornell")
    r.film_init(W, H, SEED)
    r.render(4, pa.LI_PATHSAMPLED)
    r.sync()
    want = r.film()
    free0 = torch.cuda.mem_get_info()[0]
    # 4096 x 4096 pixels x 64 samples would be 2^30 slots x 308 B x 3 lanes = 0.99 TB: must shrink (to at most 2^29 slots, then to what is free)
    old = os.environ.get("PTR_PASS_SLOTS")
    os.environ["PTR_PASS_SLOTS"] = str(1 << 40)
    try:
        r.film_init(4096, 4096, SEED)
        free1 = torch.cuda.mem_get_info()[0]
        assert free1 < free0  # something was allocated, and within the device
        r.film_init(W, H, SEED)  # back to the small film: the big buffers are released
        r.render(4, pa.LI_PATHSAMPLED)
        r.sync()
        got = r.film()
        for a, b in zip(want, got):
            assert np.array_equal(a, b)
        # a film whose pass buffers cannot exist even at one sample per pass (2^28 pixels x 308 B x 3 lanes = 248 GB, with
        # 60 % of the device taken by someone else): an error, and nothing stays allocated
        hog = torch.empty(int(free0 * 0.6), dtype=torch.uint8, device="cuda")
        try:
            with pytest.raises(pa.capi.PtrError):
                r.film_init(16384, 16384, SEED)
        finally:
            del hog
            torch.cuda.empty_cache()
        assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)
    finally:
        if old is None:
            os.environ.pop("PTR_PASS_SLOTS", None)
        else:
            os.environ["PTR_PASS_SLOTS"] = old
    r.film_init(W, H, SEED)
    r.render(4, pa.LI_PATHSAMPLED)
    r.sync()
    got = r.film()
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
