"""The N > 1 path on CPU: world_size-2 gloo ranks each take their round-robin share of the sample
indices (pathtracer_amd.dist.shard_samples - the rule ptr_render applies after ptr_film_shard),
render it (here with the oracle standing in for the GPU), all-reduce the accumulators and must end
with the film of the unsharded render."""
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
    for s in pdist.shard_samples(0, NS, rank, world):
        film = o.render(sc.camera, W, H, SEED, s, 1, pa.LI_PATHSAMPLED, 1, film=film)
    we, acc, smp = [torch.from_numpy(x) for x in film]
    pdist.all_reduce_film(d, we, acc, smp)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), we=we.numpy(), acc=acc.numpy(), smp=smp.numpy())
    d.barrier()
    d.destroy_process_group()


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
    we, acc, smp = o.render(sc.camera, W, H, SEED, 0, NS, pa.LI_PATHSAMPLED, 2)
    for r in range(2):
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert np.array_equal(g["smp"], smp)
        assert np.allclose(g["acc"], acc, rtol=1e-12)
        assert np.allclose(g["we"], we, rtol=1e-5, atol=1e-6)  # float sum order differs from the 1-rank render
