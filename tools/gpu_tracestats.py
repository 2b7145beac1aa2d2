"""Phase statistics of the one-ray-per-lane persistent tracer (dev_trace.h, PT_STATS build):
    make -C pathtracer_amd VARIANT=stats EXTRA=-DPT_STATS -j8
    PTR_TRACER=lane PTR_HIP_LIB=pathtracer_amd/lib/libptr_hip_stats.so python tools/gpu_tracestats.py
Renders passes of the bench scene; all trace launches (closest + any-hit) accumulate into one table."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa

NAMES = ["outer", "refill_it", "refill_lanes", "top_it", "top_lanes", "in_it", "in_lanes", "leaf_it", "leaf_lanes", "leaf_entry", "leaf_entry_lanes",
         "cyc_refill", "cyc_top", "cyc_in", "cyc_leaf", "cyc_total", "rays", "cyc_topprim", "topprim_it", "topprim_lanes"]
W, H = 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fam = sys.argv[2] if len(sys.argv) > 2 else "knot"
p0 = int(sys.argv[3]) if len(sys.argv) > 3 else 660
integ = int(sys.argv[4]) if len(sys.argv) > 4 else 3
sc = pa.BuiltScene(fam, W, H, p0, 0)
r = pa.Renderer(0); r.upload(sc); r.set_camera(sc.camera); r.film_init(W, H, 1)
lib = pa.capi.hip()
fn = lib.ptr_debug_trace_stats
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
buf = (C.c_uint64 * len(NAMES))()
r.render(spp, integ); r.sync()
fn(buf, len(NAMES))  # clear
r.counters_reset()
r.render(spp, integ); r.sync()
fn(buf, len(NAMES))
c = r.counters()
s = dict(zip(NAMES, [int(v) for v in buf]))
rays = c["total_paths"] + c["shadow_rays"] + c["mis_rays"]
s["rays"] = rays
out = {"raw": s}
for ph in ("refill", "top", "in", "leaf"):
    it, ln = s[ph + "_it"], s[ph + "_lanes"]
    out[ph] = {"wave_iterations_per_ray": it / rays, "lane_utilisation": ln / (64.0 * it) if it else 0, "lane_steps_per_ray": ln / rays,
               "cycle_share": s["cyc_" + ph] / max(1, s["cyc_total"])}
out["leaf"]["entries_per_ray_wave"] = s["leaf_entry"] / rays
out["leaf"]["entry_lane_utilisation"] = s["leaf_entry_lanes"] / (64.0 * max(1, s["leaf_entry"]))
out["top_primitive_tests"] = {"cycle_share": s["cyc_topprim"] / max(1, s["cyc_total"]), "wave_iterations_per_ray": s["topprim_it"] / rays,
                               "lane_utilisation": s["topprim_lanes"] / (64.0 * max(1, s["topprim_it"])), "lane_tests_per_ray": s["topprim_lanes"] / rays}
out["outer_iterations_per_ray"] = s["outer"] / rays
print(json.dumps(out, indent=1))
r.close()
