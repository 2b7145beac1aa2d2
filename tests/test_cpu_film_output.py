"""Film output without a GPU: the oracle's restatement of GetPixelRGB (camera.cpp:36-51) against
values worked out by hand in float arithmetic, and the host mirror's PNG / PPM containers
(camera.cpp:53-85, ppm.cpp:4-68) against an independent decoder (Python's zlib)."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from oracle import orc
from pathtracer_amd import capi

f32 = np.float32


def srgb(v):
    v = f32(v)
    if v <= f32(0.0031308):
        return f32(12.92) * v
    return f32(1.055) * f32(np.power(v, f32(f32(1.0) / f32(2.4)), dtype=f32)) - f32(0.055)


def by_hand(e, exposure, algo):
    e = f32(e)
    if algo == 0:
        e = e * (f32(1) / (e + f32(1)))
    elif algo == 1:
        e = f32(1) - f32(np.exp(-e * f32(exposure), dtype=f32))
    else:
        A, B, Cc, D, E, F, W = map(f32, (0.15, 0.50, 0.10, 0.20, 0.02, 0.30, 11.2))
        e = e * f32(exposure)
        e = (e * (A * e + Cc * B) + D * E) * (f32(1) / (e * (A * e + B) + D * F)) - E / F
        white = ((W * (A * W + Cc * B) + D * E) / (W * (A * W + B) + D * F)) - E / F
        e = e * (f32(1) / white)
    return int(np.float64(255.999) * np.float64(srgb(e))) & 255


@pytest.mark.parametrize("algo", [0, 1, 2])
def test_oracle_tonemap_known_answers(algo):
    rng = np.random.default_rng(5 + algo)
    n = 4096
    acc = rng.integers(1, 64, n).astype(np.float64)
    we = (rng.random((n, 3), dtype=np.float32) ** 4 * 6).astype(np.float32) * acc[:, None].astype(np.float32)
    we[:8] = 0  # black stays black
    got = orc.film_rgb8(we, acc, 2.0, algo)
    assert (got[:8] == 0).all()
    inv = (np.float32(1.0) / acc).astype(np.float32)
    e = we * inv[:, None]
    want = np.array([[by_hand(e[i, k], 2.0, algo) for k in range(3)] for i in range(n)], np.int32)
    # libm's powf/expf against numpy's: identical up to the rounding of the last place, which can move a byte by one
    assert np.abs(got.astype(np.int32) - want).max() <= 1
    assert (got.astype(np.int32) != want).mean() < 0.01


def test_oracle_tonemap_edges():
    we = np.array([[1, 2, 3], [0.5, 0.5, 0.5], [np.nan, 1, 1]], np.float32)
    acc = np.array([0.0, 1.0, 1.0])
    out = orc.film_rgb8(we, acc, 2.0, 1)
    assert (out[0] == 0).all()  # zero accWeight: defined as black at the boundary
    assert out[2, 0] == 0 and out[2, 1] > 0  # NaN -> 0
    with pytest.raises(ValueError):
        orc.film_rgb8(we, acc, 2.0, 7)


def decode_png(buf):
    assert buf[:8] == b"\x89PNG\r\n\x1a\n"
    at, chunks = 8, []
    while at < len(buf):
        n, tag = struct.unpack(">I4s", buf[at:at + 8])
        body = buf[at + 8:at + 8 + n]
        (crc,) = struct.unpack(">I", buf[at + 8 + n:at + 12 + n])
        assert crc == zlib.crc32(tag + body)
        chunks.append((tag, body))
        at += 12 + n
    assert [t for t, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype, comp, flt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, flt, lace) == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (64, 33)])
def test_png_container_round_trip(w, h):
    rng = np.random.default_rng(w * 100 + h)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    H = capi.host()
    cap = H.pth_png_encode(None, w, h, 1, None, 0)
    out = np.empty(cap, np.uint8)
    n = H.pth_png_encode(rgb.ctypes.data, w, h, 1, out.ctypes.data, cap)
    assert 0 < n <= cap
    # flip = MZ_TRUE: the bottom-up pixel array comes out top-down
    assert (decode_png(out[:n].tobytes()) == rgb[::-1]).all()
    n = H.pth_png_encode(rgb.ctypes.data, w, h, 0, out.ctypes.data, cap)
    assert (decode_png(out[:n].tobytes()) == rgb).all()
    assert H.pth_png_encode(rgb.ctypes.data, w, h, 0, out.ctypes.data, 10) == -1


def test_ppm_read_matches_the_writer_layout(tmp_path):
    w, h = 5, 3
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)  # bottom-up
    p = tmp_path / "a.ppm"
    p.write_bytes(b"P6\n%d %d\n255\n" % (w, h) + rgb[::-1].tobytes())
    out = np.empty((h, w, 3), np.uint8)
    ww, hh = C.c_int(), C.c_int()
    assert capi.host().pth_ppm_read(str(p).encode(), out.ctypes.data, out.size, C.byref(ww), C.byref(hh)) == 0
    assert (ww.value, hh.value) == (w, h)
    assert (out == rgb).all()
    assert capi.host().pth_ppm_read(str(tmp_path / "missing.ppm").encode(), None, 0, C.byref(ww), C.byref(hh)) == 1
