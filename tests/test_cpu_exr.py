"""OpenEXR ingest for environment maps (the reader half): ReadImageEXR against files produced
by an independent writer in this file (the reference ships no .exr), and InsertEXRLightMap against the same
pixels handed to InsertLightMap directly."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

import pathtracer_amd as pa
from pathtracer_amd import assets, capi

NONE, RLE, ZIPS, ZIP = 0, 1, 2, 3
UINT, HALF, FLOAT = 0, 1, 2


def rle(data):
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        j = i + 1
        while j < n and data[j] == data[i] and j - i < 128:
            j += 1
        if j - i >= 3:
            out += struct.pack("b", j - i - 1) + bytes([data[i]])
            i = j
        else:
            j = i
            while j < n and j - i < 127 and not (j + 2 < n and data[j] == data[j + 1] == data[j + 2]):
                j += 1
            out += struct.pack("b", -(j - i)) + bytes(data[i:j])
            i = j
    return bytes(out)


def filtered(raw):
    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 128 + 256) % 256
    return d.astype(np.uint8).tobytes()


def write_exr(path, channels, compression, window=(0, 0), decreasing=False):
    """channels: dict name -> (pixel type, [h, w] array); stored in alphabetical order as the format requires."""
    names = sorted(channels)
    h, w = next(iter(channels.values()))[1].shape
    x0, y0 = window

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", channels[n][0], 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)
    head = (b"\x76\x2f\x31\x01" + struct.pack("<i", 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression])) +
            attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", bytes([1 if decreasing else 0])) +
            attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) +
            attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    lpb = 16 if compression == ZIP else 1
    dt = {UINT: np.uint32, HALF: np.float16, FLOAT: np.float32}
    blocks = []
    for b0 in range(0, h, lpb):
        raw = b"".join(channels[n][1][y].astype(dt[channels[n][0]]).tobytes() for y in range(b0, min(h, b0 + lpb)) for n in names)
        if compression == NONE:
            packed = raw
        elif compression == RLE:
            packed = rle(filtered(raw))
        else:
            packed = zlib.compress(filtered(raw))
        if len(packed) >= len(raw):
            packed = raw
        blocks.append((y0 + b0, packed))
    if decreasing:
        blocks = blocks[::-1]
    table_at = len(head)
    at = table_at + 8 * len(blocks)
    offs, body = {}, b""
    for y, packed in blocks:
        offs[y] = at + len(body)
        body += struct.pack("<ii", y, len(packed)) + packed
    table = b"".join(struct.pack("<Q", offs[y]) for y in sorted(offs))  # the table is always in increasing y
    open(path, "wb").write(head + table + body)


def picture(h, w, seed):
    rng = np.random.default_rng(seed)
    img = (rng.random((h, w, 3)) ** 3 * 50).astype(np.float32)
    img[: h // 3] = 0.25  # flat region: run-length and deflate actually pack something
    return img


@pytest.mark.parametrize("compression", [NONE, RLE, ZIPS, ZIP])
@pytest.mark.parametrize("ptype", [HALF, FLOAT])
def test_exr_round_trip(tmp_path, compression, ptype):
    img = picture(37, 23, compression * 3 + ptype)
    p = tmp_path / "a.exr"
    write_exr(p, {"R": (ptype, img[..., 0]), "G": (ptype, img[..., 1]), "B": (ptype, img[..., 2]), "A": (ptype, np.ones((37, 23), np.float32))}, compression)
    got = assets.load_exr(p)
    want = img if ptype == FLOAT else img.astype(np.float16).astype(np.float32)
    assert got.shape == (37, 23, 3) and np.array_equal(got, want)  # bit-exact: half -> float is exact


def test_exr_layout_variants(tmp_path):
    img = picture(20, 9, 1)
    # no alpha, mixed pixel types, a data window that does not start at 0, blocks stored bottom-up
    p = tmp_path / "b.exr"
    write_exr(p, {"B": (HALF, img[..., 2]), "G": (FLOAT, img[..., 1]), "R": (UINT, np.floor(img[..., 0]))}, ZIP, window=(-4, 7), decreasing=True)
    got = assets.load_exr(p)
    assert np.array_equal(got[..., 0], np.floor(img[..., 0])) and np.array_equal(got[..., 1], img[..., 1])
    assert np.array_equal(got[..., 2], img[..., 2].astype(np.float16).astype(np.float32))
    # one luminance channel feeds R, G and B
    q = tmp_path / "y.exr"
    write_exr(q, {"Y": (FLOAT, img[..., 0])}, ZIPS)
    got = assets.load_exr(q)
    assert all(np.array_equal(got[..., k], img[..., 0]) for k in range(3))
    # half specials
    s = tmp_path / "s.exr"
    vals = np.array([[0.0, -0.0, 6.0e-8, 6.1e-5, 65504.0, np.inf, -np.inf, 1.0]], np.float16)
    write_exr(s, {"R": (HALF, vals), "G": (HALF, vals), "B": (HALF, vals)}, NONE)
    got = assets.load_exr(s)
    assert np.array_equal(got[..., 0].view(np.uint32), vals.astype(np.float32).view(np.uint32))
    # LoadTextureImageData takes the .exr route by extension (texture.cpp:32)
    assert np.array_equal(assets.load_image(q), assets.load_exr(q))


def test_exr_failures(tmp_path):
    img = picture(8, 8, 2)
    good = tmp_path / "g.exr"
    write_exr(good, {"R": (FLOAT, img[..., 0]), "G": (FLOAT, img[..., 1]), "B": (FLOAT, img[..., 2])}, ZIP)
    data = good.read_bytes()
    assert assets.load_exr(tmp_path / "missing.exr") is None
    (tmp_path / "cut.exr").write_bytes(data[: len(data) // 2])
    assert assets.load_exr(tmp_path / "cut.exr") is None
    (tmp_path / "magic.exr").write_bytes(b"PNG!" + data[4:])
    assert assets.load_exr(tmp_path / "magic.exr") is None
    piz = data.replace(b"compression\0compression\0" + struct.pack("<i", 1) + bytes([ZIP]), b"compression\0compression\0" + struct.pack("<i", 1) + bytes([4]))
    (tmp_path / "piz.exr").write_bytes(piz)
    assert assets.load_exr(tmp_path / "piz.exr") is None  # PIZ is reported, not mis-decoded
    tiled = data[:4] + struct.pack("<i", 2 | 0x200) + data[8:]
    (tmp_path / "tiled.exr").write_bytes(tiled)
    assert assets.load_exr(tmp_path / "tiled.exr") is None


def envmap_arrays(scene):
    fs = scene.flat.contents
    assert fs.has_envmap
    e = fs.envmap
    n = e.width * e.height
    out = {"dims": (e.width, e.height, e.nu, e.nv), "marg_int": e.marg_int,
           "texels": np.ctypeslib.as_array(e.texels, shape=(n * 3,)).copy(),
           "func": np.ctypeslib.as_array(e.func, shape=(e.nu * e.nv,)).copy(),
           "cdf": np.ctypeslib.as_array(e.cdf, shape=((e.nu + 1) * e.nv,)).copy(),
           "marg_cdf": np.ctypeslib.as_array(e.marg_cdf, shape=(e.nv + 1,)).copy()}
    return out


def test_insert_exr_light_map_equals_insert_light_map(tmp_path):
    """ZooScene with its sky read from an .exr (float, ZIP) builds the very light map it builds from the pixels."""
    w, h = 256, 128
    sky = np.empty((h, w, 3), np.float32)
    capi.host().pth_make_sky(w, h, sky.ctypes.data)
    p = tmp_path / "sky.exr"
    write_exr(p, {"R": (FLOAT, sky[..., 0]), "G": (FLOAT, sky[..., 1]), "B": (FLOAT, sky[..., 2])}, ZIP)
    a = envmap_arrays(pa.BuiltScene("zoo", 64, 48, 0, 1))
    b = envmap_arrays(pa.BuiltScene("zoo", 64, 48, 0, 1, sky_exr=p))
    assert a["dims"] == b["dims"] and a["marg_int"] == b["marg_int"]
    for k in ("texels", "func", "cdf", "marg_cdf"):
        assert np.array_equal(a[k], b[k]), k
