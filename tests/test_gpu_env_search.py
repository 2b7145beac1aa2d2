"""Light-map importance sampling, device against oracle, to the bit.

Distribution2D::SampleContinuous (sampling.h:19-77) is a search for the last CDF entry <= u followed by IEEE arithmetic only
(no libm): the device's 16-ary search trees (dev_light.h find_interval_tree, built at upload) must land on the very interval the
oracle's restatement of FindInterval lands on, for every u - including u on a CDF entry, u = 0, the largest float below 1, and
rows whose function is zero over a stretch (equal CDF entries: "the last entry <= u")."""
import numpy as np
import pytest

import pathtracer_amd as pa
from oracle import orc

pytestmark = pytest.mark.gpu


def rows_for(seed, n=120000):
    rng = np.random.default_rng(seed)
    u = rng.random((n, 2)).astype(np.float32)
    edge = np.array([0.0, np.nextafter(np.float32(0), np.float32(1)), 0.25, 0.5, 0.75, np.nextafter(np.float32(1), np.float32(0)), 1.0 / 3.0,
                     1.0 / 256.0, 1.0 / 512.0, 255.0 / 256.0, 511.0 / 512.0, 1e-7, 1 - 1e-6], dtype=np.float32)
    g = np.stack(np.meshgrid(edge, edge), -1).reshape(-1, 2)
    k = (np.arange(0, 1025, dtype=np.float32) / np.float32(1024.0)).clip(0, np.nextafter(np.float32(1), np.float32(0)))  # multiples of 1 / 1024: CDF entries of flat stretches
    kk = np.stack([k, k[::-1]], 1)
    k2 = np.stack([rng.random(k.size).astype(np.float32), k], 1)
    k3 = np.stack([k, rng.random(k.size).astype(np.float32)], 1)
    return np.ascontiguousarray(np.concatenate([u, g, kk, k2, k3]).astype(np.float32))


@pytest.mark.parametrize("fam,p0,p1", [("zoo", 0, 1), ("knot", 24, 0), ("furnace", 0, 7), ("fluid", 2000, 0), ("subsurface", 16, 0)])
def test_env_sample_and_pdf_bit_identical(cuda, fam, p0, p1):
    sc = pa.BuiltScene(fam, 32, 24, p0, p1)
    r = pa.Renderer(0)
    r.upload(sc)
    o = orc.OracleScene(sc.flat, sc.rand_calls)
    rows = rows_for(5)
    a = r.unit_eval("env_sample", rows)
    b = orc.unit_eval("env_sample", rows, o)
    assert a.shape == b.shape
    same = a.view(np.uint32) == b.view(np.uint32)
    assert same.all(), "%d of %d samples differ, first at u = %s: device %s oracle %s" % (
        (~same.all(axis=1)).sum(), rows.shape[0], rows[np.argmin(same.all(axis=1))], a[np.argmin(same.all(axis=1))], b[np.argmin(same.all(axis=1))])
    uv = np.ascontiguousarray(a[:, :2])
    pa_, pb_ = r.unit_eval("env_pdf", uv), orc.unit_eval("env_pdf", uv, o)
    assert np.array_equal(pa_.view(np.uint32), pb_.view(np.uint32))
    # the pdf the sample reports is the pdf of where it landed (sampling.h:63-69) - zero-function stretches and samples that
    # land on a cell border (the look-up then reads the neighbouring cell) aside
    ok = a[:, 2] > 0
    assert np.isclose(a[ok, 2], pa_[ok, 0], rtol=1e-4).mean() > 0.999
    o.close()
    r.close()
    sc.close()
