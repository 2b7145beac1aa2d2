"""The entry point that allocates nothing: ptr_trace_closest_noalloc (include/ptr_api.h) against ptr_trace_closest, through
the C-ABI, on two calls in a row, the second with a smaller input (the static scratch keeps what the first call left).
Collected last (test_zz_*): a failure here must not hide the tests before it."""
import ctypes as C

import numpy as np
import pytest

import pathtracer_amd as pa

pytestmark = pytest.mark.gpu
FMAX = np.finfo(np.float32).max


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-30, 30, size=(3, n)).astype(np.float32)
    a[1] = rng.uniform(0, 40, size=n).astype(np.float32)
    b = rng.uniform(-12, 12, size=(3, n)).astype(np.float32)
    b[1] += 14
    d = b - a
    d /= np.maximum(np.sqrt((d.astype(np.float64) ** 2).sum(0)), 1e-9).astype(np.float32)
    return a, d.astype(np.float32)


def _trace(r, fn, o, d, tmax, check_no_alloc=False):
    import torch

    dev = torch.device("cuda:0")
    n = o.shape[1]
    to, td, tt = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (o, d, tmax))
    out = {"prim": torch.full((n,), -9, dtype=torch.int32, device=dev), "face": torch.full((n,), -9, dtype=torch.int32, device=dev),
           "t": torch.zeros(n, device=dev), "b0": torch.zeros(n, device=dev), "b1": torch.zeros(n, device=dev)}
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    fn(n, to.data_ptr(), td.data_ptr(), tt.data_ptr(), out["prim"].data_ptr(), out["face"].data_ptr(), out["t"].data_ptr(),
       out["b0"].data_ptr(), out["b1"].data_ptr())
    r.sync()
    if check_no_alloc:
        assert torch.cuda.mem_get_info()[0] == free0  # the call allocated nothing on the device
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("tracer", ["lane", "quad"])
def test_noalloc_trace_matches(cuda, tracer, monkeypatch):
    import torch

    monkeypatch.setenv("PTR_TRACER", tracer)
    sc = pa.BuiltScene("knot", 64, 48, 40, 1)  # mesh + sphere: both levels of the traversal
    with pa.Renderer(0) as r:
        r.upload(sc)
        for n, seed in ((200_000, 5), (1_237, 6)):  # second call: smaller input, cursor left beyond it by the first
            o, d = _rays(n, seed)
            tmax = np.full(n, FMAX, np.float32)
            tmax[::7] = 25.0
            # free device memory must not move over the call; the first launch of a kernel makes the HIP runtime set up its
            # private-segment memory, which is not the library's doing, so the check is made from the second call on
            got = _trace(r, r.trace_closest_noalloc, o, d, tmax, check_no_alloc=n < 100_000)
            want = _trace(r, r.trace_closest, o, d, tmax)
            assert (want["prim"] >= 0).mean() > 0.2  # the batch does hit things
            for k in ("prim", "face"):
                assert np.array_equal(got[k], want[k]), k
            for k in ("t", "b0", "b1"):
                assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k  # bit for bit
        # beyond the documented maximum: refused before anything is touched
        rc = pa.capi.hip().ptr_trace_closest_noalloc(r._ctx, (1 << 30) + 1, None, None, None, None, None, None, None, None)
        assert rc == pa.capi.PTR_ERR_LIMIT
        assert b"PTR_TRACE_NOALLOC_MAX_RAYS" in pa.capi.hip().ptr_last_error()
        # and an empty batch is a no-op
        assert pa.capi.hip().ptr_trace_closest_noalloc(r._ctx, 0, None, None, None, None, None, None, None, None) == 0
