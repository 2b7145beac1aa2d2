"""Film output on the GPU: ptr_film_rgb8 against the oracle's GetPixelRGB, the
host mirror's ImageWrite, and checkpoint / resume of the accumulators through ptr_film_write."""
import zlib

import numpy as np
import pytest

import pathtracer_amd as pa
from oracle import orc

import common
from test_cpu_film_output import decode_png

pytestmark = pytest.mark.gpu
W, H, SEED = 64, 48, 11


@pytest.fixture(scope="module")
def rendered(cuda):
    sc = common.build("cornell_glass", W, H)
    r = pa.Renderer(0)
    r.upload(sc)
    r.set_camera(sc.camera)
    r.film_init(W, H, SEED)
    r.render(8, pa.LI_PATHSAMPLED)
    yield sc, r
    r.close()


@pytest.mark.parametrize("algo", [pa.TONEMAP_REINHARD, pa.TONEMAP_EXPONENTIAL, pa.TONEMAP_NAUGHTYDOG])
@pytest.mark.parametrize("exposure", [1.0, 2.0])
def test_rgb8_matches_the_oracle(rendered, algo, exposure):
    sc, r = rendered
    we, acc, _ = r.film()
    got = r.rgb8(exposure, algo)
    want = orc.film_rgb8(we, acc, exposure, algo)
    assert got.shape == (H, W, 3) and got.dtype == np.uint8
    assert want.max() > 100  # the scene is lit
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    d = np.minimum(d, 256 - d)  # values above 1 wrap when narrowed to a byte, in the reference too
    # tolerance: device powf / expf against libm's differ in the last place, which moves a byte whose
    # scaled value sits on an integer boundary by one; everything else is the same float arithmetic
    assert d.max() <= 1
    assert (d != 0).mean() < 0.01


def test_rgb8_synthetic_film_covers_the_value_range(rendered):
    """Accumulators written through ptr_film_write: black, sub-linear-segment values, bright values that wrap
    in NaughtyDog, zero-weight pixels."""
    sc, r = rendered
    keep = [x.copy() for x in r.film()]
    rng = np.random.default_rng(3)
    n = W * H
    acc = rng.integers(1, 50, n).astype(np.float64)
    e = np.concatenate([np.zeros(64), np.geomspace(1e-6, 40.0, n - 64)]).astype(np.float32)
    we = np.stack([e, e[::-1], rng.permutation(e)], 1) * acc[:, None].astype(np.float32)
    acc[100:110] = 0
    r.film_write(we, acc, np.ones(n, np.int64) * 3, 3)
    back = r.film()
    assert np.array_equal(back[0].reshape(n, 3), we) and np.array_equal(back[1].reshape(n), acc)
    for algo in (0, 1, 2):
        got = r.rgb8(1.5, algo).reshape(n, 3)
        want = orc.film_rgb8(we, acc, 1.5, algo)
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        d = np.minimum(d, 256 - d)  # a wrap at 255|0 is the same one-step difference
        assert d.max() <= 1, (algo, int(d.argmax()))
        assert (got[100:110] == 0).all()
    r.film_write(*keep, -1)
    with pytest.raises(pa.PtrError):
        r.rgb8(2.0, 9)


def test_image_write_png_and_ppm(rendered, tmp_path):
    sc, r = rendered
    rgb = r.rgb8(2.0, pa.TONEMAP_EXPONENTIAL)
    p = tmp_path / "output.png"
    r.save(p)  # render()'s defaults: exposure 2, Exponential (main.cpp:1185)
    assert (decode_png(p.read_bytes()) == rgb[::-1]).all()
    q = tmp_path / "backup.ppm"
    r.save(q)
    head = b"P6\n%d %d\n255\n" % (W, H)
    raw = q.read_bytes()
    assert raw.startswith(head)
    assert (np.frombuffer(raw[len(head):], np.uint8).reshape(H, W, 3) == rgb[::-1]).all()
    with pytest.raises(OSError):
        r.save(tmp_path / "no_such_dir" / "x.png")


def test_checkpoint_resume_continues_the_sample_sequence(rendered, tmp_path):
    """6 samples straight == 2 samples, checkpoint, a fresh film restored from it, 4 more."""
    sc, r = rendered
    keep = [x.copy() for x in r.film()]
    r.film_init(W, H, SEED)
    r.render(6, pa.LI_PATH)
    a = [x.copy() for x in r.film()]
    r.film_init(W, H, SEED)
    r.render(2, pa.LI_PATH)
    ck = tmp_path / "film.ckpt"
    r.checkpoint(ck)
    r.film_init(W, H, SEED)  # what a restarted process has
    r.resume(ck)
    r.render(4, pa.LI_PATH)
    b = r.film()
    assert np.array_equal(a[2], b[2]) and (b[2] == 6).all()
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[0], b[0])
    r.film_init(W, H, SEED)
    r.film_write(*keep, -1)
