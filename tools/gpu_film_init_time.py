"""how long ptr_film_init takes (pass buffers sized from free memory): PYTHONPATH=. python3 tools/gpu_film_init_time.py"""
import time
import pathtracer_amd as pa
sc = pa.BuiltScene("cornell", 64, 48, 0, 0)
with pa.Renderer(0) as r:
    r.upload(sc)
    r.set_camera(sc.camera)
    for w, h in ((64, 48), (1920, 1080), (1920, 1080), (3840, 2160), (64, 48)):
        t = time.time()
        r.film_init(w, h, 1)
        r.sync()
        print("%dx%d film_init %.3f s" % (w, h, time.time() - t), flush=True)
