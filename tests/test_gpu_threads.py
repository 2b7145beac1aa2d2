"""Two host threads, each with its own context on the same GPU, rendering different scenes at the same time (the shape of
a one-thread-per-context driver, host/pt_multigpu.cpp): the films are those of the same renders done one after the other,
bit for bit, and an error raised in one thread stays in that thread (ptr_last_error is per thread)."""
import threading

import numpy as np
import pytest

import pathtracer_amd as pa

import common

pytestmark = pytest.mark.gpu
W, H = 96, 64


_upload_lock = threading.Lock()


def render(name, integ, spp, out, key, barrier=None):
    # scene build and upload draw from the process-wide libc rand() (srand + replay in Renderer.upload, then ptr_scene_upload's
    # own draws for the top-level tree, as Aggregator::Wrap): one thread at a time, like any host that shares rand()
    with _upload_lock:
        sc = common.build(name, W, H)
        r = pa.Renderer(0)
        r.upload(sc)
    with r:
        r.set_camera(sc.camera)
        if barrier is not None:
            barrier.wait()
        for _ in range(3):  # several film inits and renders in a row, interleaving with the other thread
            r.film_init(W, H, 7)
            r.render(spp, integ)
            r.sync()
        out[key] = r.film()[0].copy()
    sc.close()


def test_two_contexts_in_two_threads(cuda):
    jobs = [("cornell_fog", pa.LI_VOLPATH, 24), ("knot_glass", pa.LI_PATHSAMPLED, 16)]
    serial, threaded, errors = {}, {}, []
    for i, (name, integ, spp) in enumerate(jobs):
        render(name, integ, spp, serial, i)
    barrier = threading.Barrier(len(jobs))

    def run(i, name, integ, spp):
        try:
            render(name, integ, spp, threaded, i, barrier)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=run, args=(i,) + j) for i, j in enumerate(jobs)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=100)
    assert not any(t.is_alive() for t in ts) and not errors, errors
    for i in range(len(jobs)):
        assert np.array_equal(serial[i], threaded[i]), jobs[i]


def test_errors_stay_in_their_thread(cuda):
    seen = {}

    def bad():
        with pa.Renderer(0) as r:
            try:
                r.render(1)  # no scene
            except pa.PtrError as e:
                seen["bad"] = str(e)

    def good():
        sc = common.build("cornell", 32, 24)
        with pa.Renderer(0) as r:
            r.upload(sc)
            r.set_camera(sc.camera)
            r.film_init(32, 24, 1)
            r.render(2, pa.LI_PATH)
            r.sync()
            seen["good"] = pa.capi.hip().ptr_last_error().decode()
        sc.close()

    a, b = threading.Thread(target=bad), threading.Thread(target=good)
    a.start(); a.join(60)
    b.start(); b.join(60)
    assert "ptr_render needs" in seen["bad"]
    assert "ptr_render needs" not in seen["good"]


def test_destroy_with_passes_in_flight(cuda):
    """ptr_render returns while its passes run; destroying the context (or replacing the film, or the scene) right away must
    wait for them, not pull their buffers away"""
    sc = common.build("knot_glass", 256, 192)
    for what in ("destroy", "film", "scene"):
        r = pa.Renderer(0)
        r.upload(sc)
        r.set_camera(sc.camera)
        r.film_init(256, 192, 3)
        r.render(256, pa.LI_VOLPATH)  # ~12 M paths over several lanes, not waited for
        if what == "film":
            r.film_init(128, 96, 3)
            r.render(4, pa.LI_PATH)
        elif what == "scene":
            r.upload(sc)
            r.render(4, pa.LI_PATH)
        r.close()
    with pa.Renderer(0) as r:  # the device is fine afterwards
        r.upload(sc)
        r.set_camera(sc.camera)
        r.film_init(64, 48, 3)
        r.render(4, pa.LI_PATHSAMPLED)
        r.sync()
        assert np.isfinite(r.image()).all()
    sc.close()
