"""Multi-GPU layout of the render path: the scene is replicated, sample indices are dealt
round-robin to the ranks (rank r owns samples s with s % world == r), and the float accumulators
(Pixel::we, Pixel::accWeight, Pixel::samples) are summed with one all-reduce - RCCL on GPUs
("nccl" backend), gloo in the CPU tests.  There is no exchange while rendering.  The reference is
single-device (cutil.cpp:84-139); this has no counterpart there."""
import os


def shard_samples(begin, ns, offset, stride):
    """Sample indices in [begin, begin+ns) owned by shard (offset, stride) - the same rule as
    ptr_render after ptr_film_shard (ptr_runtime.hip)."""
    first = begin + ((offset - begin) % stride)
    return range(first, begin + ns, stride)


TILE = 8  # PTR_TILE, include/ptr_api.h


def tile_pixels(width, height, offset, stride):
    """Film pixel indices (row-major, y * width + x) of the 8x8 tiles t with t % stride == offset, in the order
    ptr_film_tiles lays them out (tile by tile, rows of a tile top to bottom) - the same rule as the library's."""
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    out = []
    for t in range(offset, tx * ty, stride):
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        for y in range(y0, min(y0 + TILE, height)):
            out.extend(range(y * width + x0, y * width + min(x0 + TILE, width)))
    return out


class DeviceArray:
    """A raw device pointer dressed up for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def film_tensors(renderer, device):
    """torch views (no copy) of the accumulators inside the library: we[3n] f32, acc[n] f64, samples[n] i64"""
    import torch

    n = renderer.width * renderer.height
    we_p, acc_p, smp_p = renderer.film_device_ptrs()
    return (torch.as_tensor(DeviceArray(we_p, (3 * n,), "<f4"), device=device),
            torch.as_tensor(DeviceArray(acc_p, (n,), "<f8"), device=device),
            torch.as_tensor(DeviceArray(smp_p, (n,), "<i8"), device=device))


def init_process_group(backend):
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def all_reduce_film(dist, we, acc, samples=None, total_samples=None):
    """Sum the per-rank accumulators in place (every rank ends with the full film): two reductions, issued together
    (async) and waited for once.  The sample counts need no reduction when the caller knows them: after a render in
    which every pixel received `total_samples` samples over all ranks (both the tile split and the sample split give
    that), samples is simply filled; without total_samples it is reduced like the others."""
    works = [dist.all_reduce(we, async_op=True), dist.all_reduce(acc, async_op=True)]
    if samples is not None:
        if total_samples is not None:
            samples.fill_(int(total_samples))
        else:
            works.append(dist.all_reduce(samples, async_op=True))
    for w in works:
        w.wait()
