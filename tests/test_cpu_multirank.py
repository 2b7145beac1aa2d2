"""The N > 1 path on CPU: world_size-2 gloo ranks each take their round-robin share of the sample
indices (pathtracer_amd.dist.shard_samples - the rule ptr_render applies after ptr_film_shard) or their
8x8 tiles, render them (here with the oracle standing in for the GPU), assemble the film on rank 0 with
pathtracer_amd.dist.FilmGather and must end with the film of the unsharded render - after every step of
render -> gather -> render -> gather, and however often the gather is repeated (the ranks' own films are
never written by it)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SEED, NS = 24, 16, 5, 6


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


HALF = NS // 2


def _rank(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import common
    import pathtracer_amd as pa
    from pathtracer_amd import dist as pdist

    d = pdist.init_process_group("gloo")
    sc = common.build("cornell_glass", W, H)
    o = common.oracle_of(sc)
    film = (np.zeros((W * H, 3), np.float32), np.zeros(W * H, np.float64), np.zeros(W * H, np.int64))
    fg = pdist.FilmGather(d, W, H, "samples", "cpu")
    out = {}
    for step, (begin, ns) in enumerate(((0, HALF), (HALF, NS - HALF))):
        for s in pdist.shard_samples(begin, ns, rank, world):
            film = o.render(sc.camera, W, H, SEED, s, 1, pa.LI_PATHSAMPLED, 1, film=film)
        we, acc, smp = [torch.from_numpy(np.ascontiguousarray(x)).reshape(-1) for x in film]
        mine = [t.clone() for t in (we, acc, smp)]
        for rep in range(2):  # gathering twice gives the same film: nothing is summed in place
            res = fg.run(we, acc, smp, total_samples=None if rep else begin + ns)
            if rank == 0:
                for k, t in zip(("we", "acc", "smp"), res):
                    out["%s_%d_%d" % (k, step, rep)] = t.numpy().copy()
        assert all(torch.equal(a, b) for a, b in zip(mine, (we, acc, smp)))  # the rank's own film is untouched
    if rank == 0:
        np.savez(os.path.join(out_dir, "rank0.npz"), **out)
    d.barrier()
    d.destroy_process_group()


def _rank_tiles(rank, world, port, out_dir):
    """tile split: each rank keeps the pixels of its 8x8 tiles (every sample of them) and zeros elsewhere"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import common
    import pathtracer_amd as pa
    from pathtracer_amd import dist as pdist

    d = pdist.init_process_group("gloo")
    sc = common.build("cornell_glass", W, H)
    o = common.oracle_of(sc)
    mine = np.zeros(W * H, bool)
    mine[pdist.tile_pixels(W, H, rank, world)] = True
    fg = pdist.FilmGather(d, W, H, "tiles", "cpu")
    film = None
    out = {}
    for step, (begin, ns) in enumerate(((0, HALF), (HALF, NS - HALF))):
        # per-pixel streams: a pixel's film does not depend on who else is rendered
        film = o.render(sc.camera, W, H, SEED, begin, ns, pa.LI_PATHSAMPLED, 1, film=film)
        we, acc, smp = np.array(film[0]).reshape(-1, 3), np.array(film[1]), np.array(film[2])
        we[~mine] = 0
        acc[~mine] = 0
        smp[~mine] = 0
        t = [torch.from_numpy(we).reshape(-1), torch.from_numpy(acc), torch.from_numpy(smp)]
        for rep in range(2):
            res = fg.run(*t)
            if rank == 0:
                for k, x in zip(("we", "acc", "smp"), res):
                    out["%s_%d_%d" % (k, step, rep)] = x.numpy().copy()
    if rank == 0:
        np.savez(os.path.join(out_dir, "tile_rank0.npz"), **out)
    d.barrier()
    d.destroy_process_group()


def test_packed_tile_index_is_a_partition():
    """the packed layout of ptr_film_tiles_pack: every pixel in exactly one shard's packed film, at the documented entry"""
    from pathtracer_amd.dist import packed_tile_index, packed_tile_pixels, tile_pixels

    for w, h in ((24, 16), (37, 21), (8, 8), (5, 3), (64, 1)):
        for world in (1, 2, 3, 8):
            seen = np.zeros(w * h, int)
            for r in range(world):
                idx = packed_tile_index(w, h, r, world)
                assert len(idx) == packed_tile_pixels(w, h, world) and len(idx) % 64 == 0
                inside = idx[idx >= 0]
                assert sorted(inside) == sorted(tile_pixels(w, h, r, world))
                seen[inside] += 1
                tx = (w + 7) // 8
                for e in np.nonzero(idx >= 0)[0]:
                    x, y = idx[e] % w, idx[e] // w
                    assert e == (((y // 8) * tx + x // 8 - r) // world) * 64 + (y & 7) * 8 + (x & 7)
            assert (seen == 1).all()


def test_tile_pixels_partition():
    from pathtracer_amd.dist import tile_pixels

    for w, h in ((24, 16), (37, 21), (8, 8), (5, 3), (64, 1)):
        for world in (1, 2, 3, 8, 64):
            parts = [tile_pixels(w, h, r, world) for r in range(world)]
            flat = sorted(p for part in parts for p in part)
            assert flat == list(range(w * h))  # every pixel exactly once
            tx = (w + 7) // 8
            for r, part in enumerate(parts):
                for p in part:
                    assert (((p // w) // 8) * tx + (p % w) // 8) % world == r


def test_two_ranks_tile_split_is_bit_identical(tmp_path):
    """the tile split gathers disjoint tiles: the assembled film equals the unsharded one bit for bit, after each of two
    render steps and on a repeated gather"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common
    import pathtracer_amd as pa

    port = free_port()
    mp.spawn(_rank_tiles, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sc = common.build("cornell_glass", W, H)
    o = common.oracle_of(sc)
    g = np.load(os.path.join(str(tmp_path), "tile_rank0.npz"))
    film = None
    for step, (begin, ns) in enumerate(((0, HALF), (HALF, NS - HALF))):
        film = o.render(sc.camera, W, H, SEED, begin, ns, pa.LI_PATHSAMPLED, 2, film=film)
        we, acc, smp = film
        for rep in range(2):
            assert np.array_equal(g["smp_%d_%d" % (step, rep)], smp)
            assert np.array_equal(g["acc_%d_%d" % (step, rep)], acc)
            assert np.array_equal(g["we_%d_%d" % (step, rep)], np.asarray(we).reshape(-1))


def test_shard_samples_partition():
    from pathtracer_amd.dist import shard_samples

    for begin in (0, 3, 10):
        for ns in (0, 1, 5, 16):
            for world in (1, 2, 3, 8):
                got = sorted(s for r in range(world) for s in shard_samples(begin, ns, r, world))
                assert got == list(range(begin, begin + ns))
                for r in range(world):
                    assert all(s % world == r for s in shard_samples(begin, ns, r, world))


def test_two_ranks_reduce_to_the_unsharded_film(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import common
    import pathtracer_amd as pa

    port = free_port()
    mp.spawn(_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sc = common.build("cornell_glass", W, H)
    o = common.oracle_of(sc)
    g = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    film = None
    for step, (begin, ns) in enumerate(((0, HALF), (HALF, NS - HALF))):
        film = o.render(sc.camera, W, H, SEED, begin, ns, pa.LI_PATHSAMPLED, 2, film=film)
        we, acc, smp = film
        for rep in range(2):  # rep 0: sample count filled from total_samples; rep 1: reduced
            assert np.array_equal(g["smp_%d_%d" % (step, rep)], smp)
            assert np.allclose(g["acc_%d_%d" % (step, rep)], acc, rtol=1e-12)
            assert np.allclose(g["we_%d_%d" % (step, rep)], np.asarray(we).reshape(-1), rtol=1e-5, atol=1e-6)  # float sum order differs from the 1-rank render
        assert np.array_equal(g["we_%d_0" % step], g["we_%d_1" % step])  # reduce twice == reduce once
