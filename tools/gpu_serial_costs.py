"""Cost of every kernel of one pass when nothing else shares the GPU: bench.py's scene, passes
rendered one at a time (sync after each), per-family times from the library's own event timers.
Run under `rocprofv3 --kernel-trace --output-format csv` + tools/trace_timeline.py for the
per-launch view.

    python tools/gpu_serial_costs.py [--spp 8] [--passes 3] [--integrator 3] [--variant 0]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--integrator", type=int, default=3)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--segments", type=int, default=660)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    args = ap.parse_args()
    import pathtracer_amd as pa

    scene = pa.BuiltScene("knot", args.width, args.height, args.segments, args.variant)
    r = pa.Renderer(0)
    r.upload(scene)
    r.set_camera(scene.camera)
    r.film_init(args.width, args.height, 1)
    r.render(args.spp, args.integrator)
    r.sync()
    r.render(args.spp, args.integrator)  # second pass: adaptive staging has its feedback
    r.sync()
    r.profile_enable(timing=True, count_traversal=False)
    r.profile_reset()
    r.counters_reset()
    t0 = time.perf_counter()
    for _ in range(args.passes):
        r.render(args.spp, args.integrator)
        r.sync()
    dt = time.perf_counter() - t0
    c = r.counters()
    rays = c["total_paths"] + c["shadow_rays"] + c["mis_rays"] + c["probe_rays"]
    prof = r.profile()
    out = {"ms_per_pass": dt * 1e3 / args.passes, "mrays_s": rays / dt / 1e6, "rays_per_pass": rays / args.passes,
           "kernels_ms_per_pass": {k: [round(v["total_ms"] / args.passes, 3), v["launches"] // args.passes] for k, v in prof.items() if v["launches"]},
           "counters": c}
    print(json.dumps(out))
    r.close()


if __name__ == "__main__":
    main()
