"""Multi-GPU layout of the render path: the scene is replicated; either 8x8-pixel tiles are dealt round-robin to the
ranks (tile t -> rank t % world) or sample indices are (rank r owns samples s with s % world == r); the float
accumulators (Pixel::we, Pixel::accWeight, Pixel::samples) are assembled on one rank by FilmGather - a gather of
packed tiles, or a sum reduce - RCCL on GPUs ("nccl" backend), gloo in the CPU tests.  There is no exchange while
rendering.  The reference is single-device (cutil.cpp:84-139); this has no counterpart there."""
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


def packed_tile_pixels(width, height, stride):
    """entries in the packed film of each of `stride` tile shards: whole tiles of 64 pixels, sized for the largest shard
    (ptr_film_tiles_packed_pixels)"""
    tiles = ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE)
    return ((tiles + stride - 1) // stride) * TILE * TILE


def packed_tile_index(width, height, offset, stride):
    """film pixel of every entry of shard (offset, stride)'s packed film, -1 where the entry lies outside the frame: pixel
    (x, y) of the shard's k-th tile is entry k * 64 + (y & 7) * 8 + (x & 7) - the layout of ptr_film_tiles_pack, restated
    for host arrays (the gloo tests, whose films do not live in the library)."""
    import numpy as np

    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    e = np.arange(packed_tile_pixels(width, height, stride), dtype=np.int64)
    t = offset + (e >> 6) * stride
    x = (t % tx) * TILE + (e & 7)
    y = (t // tx) * TILE + ((e >> 3) & 7)
    ok = (t < tx * ty) & (x < width) & (y < height)
    return np.where(ok, y * width + x, -1)


class FilmGather:
    """The film of all ranks, assembled on rank `dst` in tensors of its own: the ranks' own films are only read, so this can
    run after every step and rendering can go on afterwards (the reference shows its film after every sample,
    main.cpp:1135-1146).
      split "tiles":   every rank packs its own 8x8 tiles (1 / world of the film, ptr_film_tiles_pack), ONE gather of the
                       packed films to dst, dst scatters them into the result: no arithmetic, the one-GPU film bit for bit;
      split "samples": the films are summed into the result with one reduce per accumulator; the sample count is not
                       reduced when the caller knows it (total_samples).
    renderer = the rank's pathtracer_amd.Renderer (packing and scattering run as kernels of the library on the film's
    device), or None for films that are plain CPU tensors (gloo tests)."""

    def __init__(self, dist, width, height, split, device, renderer=None, dst=0):
        import torch

        self.dist, self.split, self.dst, self.r = dist, split, dst, renderer
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.width, self.height = width, height
        n = width * height
        self.root = self.rank == dst
        # gloo moves CPU tensors only (the one-GPU rehearsal of N ranks, bench.py PTR_BENCH_REHEARSE_ON_GPU0): stage through the host
        self.stage = dist is not None and dist.get_backend() == "gloo" and torch.device(device).type != "cpu"
        self.we = torch.zeros(3 * n, dtype=torch.float32, device=device) if self.root or split != "tiles" else None
        self.acc = torch.zeros(n, dtype=torch.float64, device=device) if self.root or split != "tiles" else None
        self.samples = torch.zeros(n, dtype=torch.int64, device=device) if self.root else None
        if split == "tiles":
            self.entries = packed_tile_pixels(width, height, self.world)
            self.packed = torch.zeros(28 * self.entries, dtype=torch.uint8, device=device)
            self.inbox = [torch.zeros_like(self.packed) for _ in range(self.world)] if self.root else None

    def _views(self, buf):
        import torch

        e = self.entries
        return buf[:12 * e].view(torch.float32), buf[12 * e:20 * e].view(torch.float64), buf[20 * e:].view(torch.int64)

    def run(self, we, acc, samples, total_samples=None):
        """we[3n] f32, acc[n] f64, samples[n] i64: this rank's film (torch views of the library's accumulators, or CPU
        tensors).  Returns (we, acc, samples) of the whole film on rank dst, None elsewhere."""
        import torch

        dist = self.dist
        if self.split == "tiles":
            pw, pa, ps = self._views(self.packed)
            if self.r is not None:
                self.r.tiles_pack(pw.data_ptr(), pa.data_ptr(), ps.data_ptr())
                self.r.sync()  # the library's stream is not torch's: the gather below must see the packed film
            else:
                idx = torch.from_numpy(packed_tile_index(self.width, self.height, self.rank, self.world))
                ok, at = idx >= 0, idx.clamp(min=0)
                pw.view(-1, 3)[:] = torch.where(ok[:, None], we.view(-1, 3)[at], torch.zeros((), dtype=we.dtype))
                pa[:] = torch.where(ok, acc[at], torch.zeros((), dtype=acc.dtype))
                ps[:] = torch.where(ok, samples[at], torch.zeros((), dtype=samples.dtype))
            if dist is not None and self.world > 1:
                if self.stage:
                    host = [torch.zeros(self.packed.shape, dtype=torch.uint8) for _ in range(self.world)] if self.root else None
                    dist.gather(self.packed.cpu(), host, dst=self.dst)
                    if self.root:
                        for k in range(self.world):
                            self.inbox[k].copy_(host[k])
                else:
                    dist.gather(self.packed, self.inbox if self.root else None, dst=self.dst)
            elif self.root:
                self.inbox[0].copy_(self.packed)
            if not self.root:
                return None
            for k in range(self.world):
                qw, qa, qs = self._views(self.inbox[k])
                if self.r is not None:
                    torch.cuda.synchronize()
                    self.r.tiles_unpack(k, self.world, qw.data_ptr(), qa.data_ptr(), qs.data_ptr(), self.we.data_ptr(), self.acc.data_ptr(),
                                        self.samples.data_ptr())
                else:
                    idx = torch.from_numpy(packed_tile_index(self.width, self.height, k, self.world))
                    ok = idx >= 0
                    self.we.view(-1, 3)[idx[ok]] = qw.view(-1, 3)[ok]
                    self.acc[idx[ok]] = qa[ok]
                    self.samples[idx[ok]] = qs[ok]
            if self.r is not None:
                self.r.sync()
            return self.we, self.acc, self.samples
        # sample split: sum into the result, never into the film
        self.we.copy_(we)
        self.acc.copy_(acc)
        works = []
        if dist is not None and self.world > 1 and self.stage:
            hw, ha = self.we.cpu(), self.acc.cpu()
            dist.reduce(hw, dst=self.dst)
            dist.reduce(ha, dst=self.dst)
            self.we.copy_(hw)
            self.acc.copy_(ha)
        elif dist is not None and self.world > 1:
            works = [dist.reduce(self.we, dst=self.dst, async_op=True), dist.reduce(self.acc, dst=self.dst, async_op=True)]
        if self.root:
            if total_samples is not None:
                self.samples.fill_(int(total_samples))
            else:
                self.samples.copy_(samples)
        if total_samples is None and dist is not None and self.world > 1:
            tmp = samples.cpu() if self.stage else samples.clone()
            works.append(dist.reduce(tmp, dst=self.dst, async_op=True))
            for w in works:
                w.wait()
            if self.root:
                self.samples.copy_(tmp)
            works = []
        for w in works:
            w.wait()
        return (self.we, self.acc, self.samples) if self.root else None


def all_reduce_film(dist, we, acc, samples=None, total_samples=None):
    """DESTRUCTIVE, kept for callers that want the full film on every rank exactly once at the end of a render: a second
    call, or rendering on after it, counts every sample world times.  Use FilmGather for anything that looks at the film
    more than once.
    Sum the per-rank accumulators in place (every rank ends with the full film): two reductions, issued together
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
