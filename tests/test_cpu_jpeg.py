"""JPEG textures through LoadTextureImageData (host loader, pt_loaders.cpp) - SURVEY section 8(f) row 1, asset ingest.

PARITY UNPINNED against the reference's decoder (the stb_image it vendors): the reference ships no .jpg and no decoded
pixels, and a JPEG's pixels depend on the decoder's fixed-point choices.  What holds the decoder here:
  * chroma upsampling + colour conversion against numpy statements of the fixed-point formulas, EXACTLY, on files whose
    blocks are all flat (there the inverse DCT leaves no freedom, so expected bytes follow from the file's own planes);
  * the inverse DCT (unit hook) against the real-valued transform (+-1, mean error that of rounding), its DC closed form
    and transposition;
  * entropy decoding: progressive (spectral selection + successive approximation), optimised Huffman tables and
    restart intervals must give the SAME bytes as the baseline file of the same picture (same coefficients);
  * whole pictures against Pillow's decoder (libjpeg: another IDCT rounding, another upsampler) within a few levels.
Files are written by Pillow at test time; nothing of the reference is read.
"""
import ctypes as C
import io

import numpy as np
import pytest

from pathtracer_amd import assets, capi

PIL = pytest.importorskip("PIL.Image")


def decode(path):
    img = assets.load_image(path)
    assert img is not None, path
    return np.rint(img * 255.0).astype(np.int32)


def picture(w, h, seed=0):
    """smooth colour ramps + a few hard edges + mild noise: every coefficient band gets used"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    r = 127 + 100 * np.sin(x / 9.0) * np.cos(y / 13.0)
    g = 255 * (x / max(w - 1, 1))
    b = 255 * ((x // 7 + y // 5) % 2)
    img = np.stack([r, g, b], -1) + rng.normal(0, 6, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def save(tmp_path, name, arr, mode="RGB", **kw):
    p = tmp_path / name
    PIL.fromarray(arr, mode).save(p, "JPEG", **kw)
    return p


# ---- numpy restatements of the documented fixed-point arithmetic ------------------------------------------------
def upsample_expected(plane, hs, vs, w, h):
    """plane: low-resolution chroma [ph, pw] -> [h, w] by the 3:1 filters (factor 2) or repetition"""
    p = plane.astype(np.int64)
    ph, pw = p.shape
    if vs == 2:  # output row j: source row j // 2 is the nearer, the farther is the one before (even j) or after (odd j), clamped
        near = np.repeat(np.arange(ph), 2)[:h]
        far = np.clip(near + np.where(np.arange(h) % 2 == 0, -1, 1)[: len(near)], 0, ph - 1)
        rows4 = 3 * p[near] + p[far]  # times four
    else:
        rows4 = 4 * p[np.minimum(np.arange(h) // vs, ph - 1)]
    if hs == 2:
        near = np.repeat(np.arange(pw), 2)
        far = np.clip(near + np.where(np.arange(2 * pw) % 2 == 0, -1, 1), 0, pw - 1)
        if vs == 2:
            out = (3 * rows4[:, near] + rows4[:, far] + 8) >> 4
            out[:, 0] = (rows4[:, 0] + 2) >> 2  # the two ends are not interpolated sideways
            out[:, 2 * pw - 1] = (rows4[:, pw - 1] + 2) >> 2
        else:
            q = rows4 // 4
            out = (3 * q[:, near] + q[:, far] + 2) >> 2
            out[:, 0] = q[:, 0]
            out[:, 2 * pw - 1] = q[:, pw - 1]
    else:
        q = (rows4 + 2) >> 2 if vs == 2 else rows4 // 4
        out = np.repeat(q, hs, axis=1)
    return out[:, :w]


def ycc_to_rgb_expected(y, cb, cr):
    def fx(x):
        return int(np.float32(x) * np.float32(4096.0) + np.float32(0.5)) << 8

    yf = (y.astype(np.int64) << 20) + (1 << 19)
    cb = cb.astype(np.int64) - 128
    cr = cr.astype(np.int64) - 128
    r = yf + cr * fx(1.40200)
    gcb = (cb * -fx(0.34414)).astype(np.int32).view(np.uint32) & np.uint32(0xFFFF0000)
    g = yf + cr * -fx(0.71414) + gcb.view(np.int32).astype(np.int64)
    b = yf + cb * fx(1.77200)
    return np.clip(np.stack([r >> 20, g >> 20, b >> 20], -1), 0, 255)


# ---- tests ------------------------------------------------------------------------------------------------------
def test_idct_against_the_real_transform_and_closed_forms():
    """the decoder's fixed-point inverse DCT (unit hook) against the real-valued transform: never more than one level away
    where nothing clamps; a block with only a DC term is 128 + round(d / 8) everywhere (two guard bits, then 17 bits removed
    with rounding); a transposed block gives the transposed picture to within the one rounding that changes place"""
    rng = np.random.default_rng(3)
    fn = capi.host().pth_jpeg_idct
    k = np.arange(8)
    basis = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * np.where(k == 0, np.sqrt(0.5), 1.0)[:, None] * 0.5  # [u, x]

    def run(blk):
        out = np.zeros(64, np.uint8)
        b = np.ascontiguousarray(blk, np.int16)
        fn(b.ctypes.data_as(C.POINTER(C.c_short)), out.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return out.reshape(8, 8).astype(np.int64)

    worst, total, n = 0.0, 0.0, 0
    for case in range(400):
        blk = np.zeros((8, 8), np.int16)
        kind = case % 4
        if kind == 0:  # DC only
            blk[0, 0] = rng.integers(-1024, 1024)
        elif kind == 1:  # a few low frequencies
            blk[:3, :3] = rng.integers(-200, 200, (3, 3))
        elif kind == 2:  # dense, decaying
            blk[:] = (rng.normal(0, 1, (8, 8)) * 300 / (1 + k[:, None] + k[None, :]) ** 1.5).astype(np.int16)
        else:  # the first column has no AC term (the decoder's shortcut for such columns), the others do
            blk[:, 1:] = rng.integers(-60, 60, (8, 7))
            blk[0, 0] = rng.integers(-500, 500)
        got = run(blk)
        real = basis.T @ blk.astype(np.float64) @ basis + 128.0
        inside = (real > 1) & (real < 254)
        err = np.abs(got - real)[inside]
        worst = max(worst, float(err.max(initial=0)))
        total += float(err.sum())
        n += int(inside.sum())
        assert np.all(got[~inside & (real <= 1)] <= 2) and np.all(got[~inside & (real >= 254)] >= 253)
        if kind == 0:
            assert np.all(got == np.clip(128 + ((int(blk[0, 0]) + 4) >> 3), 0, 255))
        assert np.abs(run(blk.T) - got.T).max() <= 1  # columns first, rows second: transposing moves a rounding, no more
    assert worst <= 1.0 and total / n < 0.3, (worst, total / n)  # (rounding to a level alone averages 0.25)


@pytest.mark.parametrize("subsampling,hs,vs", [(0, 1, 1), (1, 2, 1), (2, 2, 2)])
@pytest.mark.parametrize("size", [(64, 48), (50, 37), (17, 16), (1, 1), (2, 3)])
def test_flat_blocks_exact(tmp_path, subsampling, hs, vs, size):
    """16x16 tiles of constant Y / Cb / Cr at quality 100 (all quantisers 1): every block is its DC term and decodes to the
    tile's value in any correct decoder, so the expected RGB follows from the planes: upsampling and colour conversion exactly"""
    w, h = size
    rng = np.random.default_rng(w * 100 + h + subsampling)
    th, tw = (h + 15) // 16, (w + 15) // 16
    tiles = rng.integers(0, 256, (3, th, tw)).astype(np.uint8)
    full = np.repeat(np.repeat(tiles, 16, axis=1), 16, axis=2)[:, :h, :w]
    im = PIL.merge("YCbCr", [PIL.fromarray(np.ascontiguousarray(full[c]), "L") for c in range(3)])
    p = tmp_path / "flat.jpg"
    im.save(p, "JPEG", quality=100, subsampling=subsampling)
    got = decode(p)
    assert got.shape == (h, w, 3)
    # the planes the file holds: luma at full size, chroma one sample per hs x vs pixels (tile values, block padding irrelevant)
    pw, ph = (w + hs - 1) // hs, (h + vs - 1) // vs
    cb = np.repeat(np.repeat(tiles[1], 16 // vs, axis=0), 16 // hs, axis=1)[:ph, :pw]
    cr = np.repeat(np.repeat(tiles[2], 16 // vs, axis=0), 16 // hs, axis=1)[:ph, :pw]
    exp = ycc_to_rgb_expected(full[0], upsample_expected(cb, hs, vs, w, h), upsample_expected(cr, hs, vs, w, h))
    assert np.array_equal(got, exp), np.abs(got - exp).max()


def test_greyscale_quality_100_is_lossless_on_flat_images(tmp_path):
    for c in (0, 1, 77, 128, 200, 255):
        p = save(tmp_path, "g%d.jpg" % c, np.full((24, 40), c, np.uint8), "L", quality=100)
        got = decode(p)
        assert got.shape == (24, 40, 3)
        assert np.all(got[..., 0] == c) and np.all(got[..., 1:] == 0)  # one channel: r = byte / 255, g = b = 0 (texture.cpp:44-53)


@pytest.mark.parametrize("subsampling", [0, 1, 2])
@pytest.mark.parametrize("size", [(96, 64), (67, 45)])
def test_entropy_coding_variants_decode_to_the_same_bytes(tmp_path, subsampling, size):
    w, h = size
    arr = picture(w, h, seed=subsampling)
    base = decode(save(tmp_path, "base.jpg", arr, quality=85, subsampling=subsampling))
    variants = {
        "progressive": dict(progressive=True),
        "optimised": dict(optimize=True),
        "restart_rows": dict(restart_marker_rows=1),
        "restart_blocks": dict(restart_marker_blocks=3),
        "progressive_restart": dict(progressive=True, restart_marker_rows=1),
    }
    for name, kw in variants.items():
        p = save(tmp_path, name + ".jpg", arr, quality=85, subsampling=subsampling, **kw)
        data = p.read_bytes()
        if name.startswith("progressive"):
            assert b"\xff\xc2" in data
        if "restart" in name:
            assert b"\xff\xdd" in data
        got = decode(p)
        assert np.array_equal(got, base), (name, np.abs(got - base).max())


@pytest.mark.parametrize("quality,subsampling,tol_max,tol_mean", [(95, 0, 3, 0.5), (85, 0, 3, 0.5), (85, 1, 12, 0.9), (85, 2, 16, 1.0), (40, 2, 16, 1.0)])
def test_against_pillow_decoder(tmp_path, quality, subsampling, tol_max, tol_mean):
    """another correct decoder (libjpeg: 13-bit IDCT constants, its own 'fancy' upsampling and colour tables): luma-only
    differences are +-1, chroma upsampling differs by a rounding convention at sharp chroma edges"""
    arr = picture(120, 88, seed=quality)
    p = save(tmp_path, "pic.jpg", arr, quality=quality, subsampling=subsampling)
    got = decode(p)
    ref = np.asarray(PIL.open(p).convert("RGB")).astype(np.int32)
    d = np.abs(got - ref)
    assert d.max() <= tol_max and d.mean() <= tol_mean, (d.max(), d.mean())
    if subsampling == 0:  # and the picture survived (the blue checker of picture() does not survive chroma subsampling in any decoder)
        mse = np.mean((got - arr.astype(np.int32)) ** 2.0)
        assert 10 * np.log10(255.0 ** 2 / mse) > 27


def test_greyscale_against_pillow(tmp_path):
    arr = picture(75, 51, seed=9)[..., 0]
    for kw in (dict(quality=90), dict(quality=90, progressive=True), dict(quality=30)):
        p = save(tmp_path, "grey.jpg", arr, "L", **kw)
        got = decode(p)[..., 0]
        ref = np.asarray(PIL.open(p)).astype(np.int32)
        assert np.abs(got - ref).max() <= 1


def test_broken_files_are_refused(tmp_path, capfd):
    arr = picture(40, 32)
    good = save(tmp_path, "ok.jpg", arr, quality=80).read_bytes()
    cases = {
        "empty.jpg": b"",
        "soi_only.jpg": good[:2],
        "no_eoi_cut_in_header.jpg": good[:30],
        "not_jpeg.jpg": b"\x00\x01\x02\x03" * 50,
        "twelve_bit.jpg": good.replace(b"\xff\xc0\x00\x11\x08", b"\xff\xc0\x00\x11\x0c"),
        "arithmetic.jpg": good.replace(b"\xff\xc0", b"\xff\xc9"),
    }
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        assert assets.load_image(p) is None, name
    # a scan cut short is not an error in the reference's decoder either (zeros are fed): the rows decoded before the cut stand
    big = save(tmp_path, "big.jpg", picture(64, 64), quality=80)
    data = big.read_bytes()
    p = tmp_path / "cut.jpg"
    p.write_bytes(data[: len(data) * 3 // 4] + b"\xff\xd9")
    got = assets.load_image(p)
    assert got is not None and got.shape == (64, 64, 3)
    assert np.array_equal(np.rint(got[:8] * 255).astype(np.int32), decode(big)[:8])


def test_jpeg_texture_reaches_a_scene(tmp_path):
    """a .jpg named by an MTL map_Kd goes through the same LoadTextureImageData as the PNG route"""
    arr = picture(32, 32, seed=5)
    jp = save(tmp_path, "tex.jpg", arr, quality=90)
    PIL.fromarray(np.asarray(PIL.open(jp).convert("RGB"))).save(tmp_path / "tex.png")
    a, b = assets.load_image(jp), assets.load_image(tmp_path / "tex.png")
    assert a.shape == b.shape == (32, 32, 3)
    assert np.abs(a - b).max() <= 3.5 / 255  # our decode vs Pillow's decode of the same file, stored losslessly
