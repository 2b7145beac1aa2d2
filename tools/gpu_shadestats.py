"""Where a shaded vertex spends its wave cycles (k_shade<false>, -DSH_STATS build):
    make -C pathtracer_amd VARIANT=shstats EXTRA=-DSH_STATS -j8
    PTR_HIP_LIB=pathtracer_amd/lib/libptr_hip_shstats.so python tools/gpu_shadestats.py [spp] [config]"""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pa

NAMES = ["load", "medium", "make_si", "scatter", "light_sample", "bsdf_eval", "nee_put_a", "bsdf_sample_mis", "light_pdf", "nee_put_b", "bsdf_sample",
         "tail", "store", "wave_rounds"]
import bench
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fam, p0, p1, W, H = bench.CONFIGS[int(sys.argv[2]) if len(sys.argv) > 2 else 1][:5]
sc = pa.BuiltScene(fam, W, H, p0, p1)
r = pa.Renderer(0); r.upload(sc); r.set_camera(sc.camera); r.film_init(W, H, 1)
fn = pa.capi.hip().ptr_debug_shade_stats
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
buf = (C.c_uint64 * len(NAMES))()
r.render(spp, 3); r.sync()
fn(buf, len(NAMES))
r.counters_reset()
r.render(spp, 3); r.sync()
fn(buf, len(NAMES))
s = dict(zip(NAMES, [int(v) for v in buf]))
tot = sum(v for k, v in s.items() if k != "wave_rounds")
print(json.dumps({"cycles_per_wave_round": {k: round(v / max(1, s["wave_rounds"]), 1) for k, v in s.items() if k != "wave_rounds"},
                  "share": {k: round(v / tot, 4) for k, v in s.items() if k != "wave_rounds"}, "wave_rounds": s["wave_rounds"],
                  "total_cycles_per_wave_round": round(tot / max(1, s["wave_rounds"]), 1)}, indent=1))
r.close()
