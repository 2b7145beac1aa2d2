"""Multi-GPU layout on the one GPU a test box has: the tile split through the C-ABI (two contexts on one device must
tile the film exactly - bit for bit the unsharded render), and the C++ RCCL driver (host/pt_multigpu.cpp) with a group
of one GPU, whose reduced film must be the single-context film."""
import ctypes as C

import numpy as np
import pytest

import pathtracer_amd as pa
from pathtracer_amd import dist as pdist

import common

pytestmark = pytest.mark.gpu
SEED, NS = 9, 5


def _render(sc, w, h, tiles=None, shard=None, integ=pa.LI_PATHSAMPLED):
    with pa.Renderer(0) as r:
        r.upload(sc)
        r.set_camera(sc.camera)
        r.film_init(w, h, SEED)
        if tiles:
            r.tiles(*tiles)
        if shard:
            r.shard(*shard)
        r.render(NS, integ)
        r.sync()
        return [a.copy() for a in r.film()]


@pytest.mark.parametrize("size", [(32, 24), (37, 21)])
@pytest.mark.parametrize("world", [2, 3])
def test_tile_shards_tile_the_film(cuda, size, world):
    w, h = size
    sc = common.build("cornell_glass", w, h)
    full = _render(sc, w, h)
    parts = [_render(sc, w, h, tiles=(k, world)) for k in range(world)]
    for k, (we, acc, smp) in enumerate(parts):
        mine = np.zeros(w * h, bool)
        mine[pdist.tile_pixels(w, h, k, world)] = True
        assert (smp.reshape(-1)[mine] == NS).all() and (smp.reshape(-1)[~mine] == 0).all()  # its own tiles and nothing else
        assert not we.reshape(-1, 3)[~mine].any() and not acc.reshape(-1)[~mine].any()
    for i in range(3):
        total = parts[0][i].copy()
        for p in parts[1:]:
            total += p[i]
        assert np.array_equal(total, full[i])  # x + 0 = x: bit for bit the unsharded film


def test_tiles_combine_with_sample_shards(cuda):
    w, h = 32, 24
    sc = common.build("cornell", w, h)
    full = _render(sc, w, h)
    parts = [_render(sc, w, h, tiles=(t, 2), shard=(s, 2)) for t in range(2) for s in range(2)]
    smp = sum(p[2] for p in parts)
    assert (smp == NS).all()
    acc = sum(p[1] for p in parts)
    assert np.allclose(acc, full[1], rtol=1e-12)
    we = sum(p[0].astype(np.float64) for p in parts)
    assert np.allclose(we, full[0], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("size", [(32, 24), (37, 21)])
@pytest.mark.parametrize("world", [2, 3])
def test_tile_gather_of_shards_on_one_device(cuda, size, world):
    """What RenderGroupReduce / FilmGather do for a tile split, emulated with `world` contexts on one device: every shard
    packs its own tiles (ptr_film_tiles_pack), shard 0 scatters all packed films into a result film of its own
    (ptr_film_tiles_unpack).  render -> gather -> gather -> render -> gather: the result equals the unsharded film bit for
    bit every time, and no shard's own film is changed by gathering."""
    import torch

    w, h = size
    sc = common.build("cornell_glass", w, h)
    dev = torch.device("cuda:0")
    rs = [pa.Renderer(0) for _ in range(world)]
    ref = pa.Renderer(0)
    try:
        for k, r in enumerate(rs + [ref]):
            r.upload(sc)
            r.set_camera(sc.camera)
            r.film_init(w, h, SEED)
            if r is not ref:
                r.tiles(k, world)
        n = rs[0].tiles_packed_pixels(world)
        assert n == pdist.packed_tile_pixels(w, h, world)
        packed = [(torch.zeros(3 * n, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.float64, device=dev),
                   torch.zeros(n, dtype=torch.int64, device=dev)) for _ in range(world)]
        res = (torch.full((3 * w * h,), -1.0, dtype=torch.float32, device=dev), torch.full((w * h,), -1.0, dtype=torch.float64, device=dev),
               torch.full((w * h,), -1, dtype=torch.int64, device=dev))
        torch.cuda.synchronize()

        def gather():
            for k, r in enumerate(rs):
                r.tiles_pack(*[t.data_ptr() for t in packed[k]])
                r.sync()
            for k in range(world):
                rs[0].tiles_unpack(k, world, *[t.data_ptr() for t in packed[k]], *[t.data_ptr() for t in res])
            rs[0].sync()
            return [t.cpu().numpy() for t in res]

        for step, ns in enumerate((2, 3)):
            for r in rs + [ref]:
                r.render(ns, pa.LI_PATHSAMPLED)
                r.sync()
            own = [[a.copy() for a in r.film()] for r in rs]
            full = ref.film()
            for rep in range(2):
                got = gather()
                for i in range(3):
                    assert np.array_equal(got[i].reshape(-1), full[i].reshape(-1)), (step, rep, i)
            # the packed layout is the documented one (dist.packed_tile_index restates it)
            for k in range(world):
                idx = pdist.packed_tile_index(w, h, k, world)
                pw = packed[k][0].cpu().numpy().reshape(-1, 3)
                assert np.array_equal(pw[idx >= 0], own[k][0].reshape(-1, 3)[idx[idx >= 0]]) and not pw[idx < 0].any()
            for r, o in zip(rs, own):
                assert all(np.array_equal(a, b) for a, b in zip(r.film(), o))  # gathering does not touch a shard's film
    finally:
        for r in rs + [ref]:
            r.close()


@pytest.mark.parametrize("split", [0, 1])
def test_render_group_of_one_gpu(cuda, split):
    """libpt_multigpu.so: ncclCommInitAll + the gather / reduce into the group's result film, with one GPU in the group.
    Reduce twice == reduce once, and render -> reduce -> render -> reduce equals the single-context film at each step."""
    w, h = 48, 32
    sc = common.build("zoo", w, h)
    m = pa.capi.multigpu()
    g = m.RenderGroupCreate(1, None, split)
    assert g, m.RenderGroupLastError()
    we = np.zeros((w * h, 3), np.float32)
    acc = np.zeros(w * h, np.float64)
    smp = np.zeros(w * h, np.int64)
    try:
        assert m.RenderGroupSize(g) == 1
        C.CDLL(None).srand(1)  # the scene build draws from libc rand() like Aggregator::Wrap; Renderer.upload reseeds the same way
        for rc in (m.RenderGroupUpload(g, sc.flat), m.RenderGroupCamera(g, C.byref(sc.camera)), m.RenderGroupFilmInit(g, w, h, SEED)):
            assert rc == 0, m.RenderGroupLastError()
        assert m.RenderGroupFilmRead(g, we.ctypes.data, acc.ctypes.data, smp.ctypes.data) != 0  # nothing reduced yet
        with pa.Renderer(0) as r:
            r.upload(sc)
            r.set_camera(sc.camera)
            r.film_init(w, h, SEED)
            for ns in (2, NS - 2):
                r.render(ns, pa.LI_VOLPATH)
                full = [a.copy() for a in r.film()]
                assert m.RenderGroupRenderSamples(g, ns, pa.LI_VOLPATH) == 0, m.RenderGroupLastError()
                for rep in range(2):
                    assert m.RenderGroupReduce(g) == 0, m.RenderGroupLastError()
                    assert m.RenderGroupFilmRead(g, we.ctypes.data, acc.ctypes.data, smp.ctypes.data) == 0, m.RenderGroupLastError()
                    assert np.array_equal(smp, full[2].reshape(-1))
                    assert np.array_equal(acc, full[1].reshape(-1))
                    assert np.array_equal(we.reshape(-1), full[0].reshape(-1))
    finally:
        m.RenderGroupFree(g)


def test_bench_multi_rank_path_over_rccl_with_one_rank(cuda):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, RCCL process group, barriers, the film's
    all-reduce on the library's own device memory, the reduction of the statistics), rehearsed with a world of one rank:
    --force-dist takes every N > 1 branch"""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "1", "--config", "0", "--steps", "2", "--warmup", "1", "--no-cpu", "--force-dist"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=250)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 100 and d["scaling"] == "weak"
    assert d["config"]["config_index"] == 0


@pytest.mark.parametrize("extra", [[], ["--scaling", "strong", "--split", "tiles"]], ids=["weak_samples", "strong_tiles"])
def test_bench_two_ranks_rehearsed_on_one_gpu(cuda, extra):
    """two ranks as torch.distributed.run starts them, both rendering on GPU 0, collectives over gloo
    (PTR_BENCH_REHEARSE_ON_GPU0: RCCL wants a GPU per rank): the rank > 0 branches of bench.py - sample / tile shards,
    reductions, only rank 0 prints - with the whole-job figure counted over both ranks"""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PTR_BENCH_REHEARSE_ON_GPU0="1", PTR_PASS_SLOTS="4000000")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--config", "0", "--steps", "2", "--warmup", "1", "--no-cpu"] + extra
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=250, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 100
    assert d["scaling"] == ("strong" if extra else "weak")
