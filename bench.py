#!/usr/bin/env python3
"""Headline benchmark of the per-ray hot path (BASELINE.json: Mrays/s, incoherent bounces).

Workload at N=1 (BASELINE.json configs[1]): 871,200-triangle torus-knot mesh ("dragon-class"
stand-in, the author's assets are not available), rough-glass microfacet material, light-map
sky, 1920x1080, Li_VolPath (the reference's default integrator, main.cpp:447).  One *step* is one
batch of --spp (16) samples of every pixel - 1/16 of the 256 spp of configs[1] - through the wavefront pipeline (raygen -> {trace, shade, nee} per
bounce -> film); scene, BVH and film are resident in HBM before the timed region starts.

value = (closest-hit + any-hit scene queries of all ranks) / wall time, in Mrays/s.
N > 1: the scene is replicated, sample indices are dealt round-robin to the ranks (rank r renders
samples r, r+N, ...: weak scaling), and the float accumulators are summed with one RCCL
all-reduce inside the timed region.

    python bench.py --gpus 1 --steps 8 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s


def ray_count(c):
    return c["total_paths"] + c["shadow_rays"] + c["mis_rays"] + c["probe_rays"]


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the newest committed PMC summary
    (profiles/r*_summary.json, written by tools/gpu_profile.sh -> tools/prof_summary.py --json from
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command).  rocprofv3 reports KB;
    on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, hence the factor 2 (MI355X_MICROARCH.md, HBM section).
    Returns (bytes_per_launch, source) or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return None, None
    try:
        k = json.load(open(files[-1])).get(kernel)
        c = k["counters"]
        n = max(1, int(k.get("dispatches") or k.get("calls") or 1))
        return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / n, os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def cpu_baseline(scene, width, height, integrator, seed):
    """The oracle (CPU restatement of the reference path) on this box's host cores, bounded sample."""
    from oracle import orc

    cores = max(1, min(16, os.cpu_count() or 1))
    osc = orc.OracleScene(scene.flat, scene.rand_calls)
    # a reduced-resolution frame of the SAME scene/camera/integrator, repeated until ~12 s of CPU work
    w, h = width // 2, height // 2
    dt, spp = 0.0, 0
    while dt < 12.0 and spp < 512:
        ns = 1 if spp == 0 else int(min(64, max(1, (12.5 - dt) / max(dt / spp, 1e-3))))
        t0 = time.time()
        osc.render(scene.camera, w, h, seed, spp, ns, integrator, cores)
        dt += time.time() - t0
        spp += ns
    rays = ray_count(osc.counters())
    osc.close()
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%dx%d x %d spp of the same scene/camera/integrator, %.1f s, %d rays" % (w, h, spp, dt, rays)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp", type=int, default=32, help="samples per pixel per step and per GPU (one batch)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--segments", type=int, default=660, help="knot tessellation: 2*s*s triangles (660 -> 871,200)")
    ap.add_argument("--variant", type=int, default=0, help="0 rough glass, 1 plastic, 2 specular glass, 3 matte")
    ap.add_argument("--integrator", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--roofline-only", action="store_true",
                    help="skip the timed region: only passes that run one at a time (what profiles/*_exclusive_* are rocprofv3 traces of)")
    args = ap.parse_args()

    import torch
    import pathtracer_amd as pa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    from pathtracer_amd import dist as pdist

    dist = None
    if world > 1:
        dist = pdist.init_process_group("nccl")
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    W, H = args.width, args.height
    scene = pa.BuiltScene("knot", W, H, args.segments, args.variant)
    r = pa.Renderer(local_rank)
    t_up = time.time()
    r.upload(scene)
    t_up = time.time() - t_up
    r.set_camera(scene.camera)
    r.film_init(W, H, args.seed)
    r.shard(rank, world)
    bvh = r.bvh_info()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        # one batch on every rank: args.spp samples of every pixel (rank r takes the sample indices = r mod world)
        r.render(world * args.spp, args.integrator)

    if args.roofline_only:
        # The timed region is skipped and every pass runs alone (sync after each), so that a kernel trace of this mode
        # (profiles/*_exclusive_*) holds nothing but launches that had the GPU to themselves.  The library's adaptive
        # staging depth gets the passes it would have seen in the timed region, so the launches are those of a full run.
        for _ in range(args.warmup + args.steps):
            step()
            r.sync()
    else:
        for _ in range(args.warmup):
            step()
        r.sync()

    # untimed counting pass: child boxes / leaf items tested per traced ray (algorithmic bytes)
    r.counters_reset()
    r.profile_enable(timing=False, count_traversal=True)
    r.profile_reset()
    step()
    r.sync()
    cc = r.counters()
    r.profile_enable(timing=True, count_traversal=False)
    r.profile_reset()
    r.counters_reset()
    r.film_init(W, H, args.seed)  # the timed region starts from a clean film
    r.shard(rank, world)

    we_t, acc_t, smp_t = pdist.film_tensors(r, dev)

    barrier()
    t0 = time.perf_counter()
    for _ in range(0 if args.roofline_only else args.steps):
        step()
    r.sync()
    if dist is not None:
        pdist.all_reduce_film(dist, we_t, acc_t, smp_t)
    barrier()
    dt = time.perf_counter() - t0

    c = r.counters()
    prof = r.profile()
    rays_local = ray_count(c)

    # roofline leg (untimed, rank 0's GPU only needs it but every rank runs it to stay in step): passes rendered one
    # at a time, so each launch has the GPU to itself - the condition rocprofv3 --kernel-trace measures under (it
    # serialises dispatches), whereas in the timed region six passes overlap and every launch shares the GPU.
    r.profile_reset()
    r.counters_reset()
    for _ in range(2):
        step()
        r.sync()
    cx = r.counters()
    profx = r.profile()

    stats = torch.tensor([dt, float(rays_local), float(c["camera_rays"])], dtype=torch.float64, device=dev)
    if dist is not None:
        tmax = stats[0:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats[1:], op=dist.ReduceOp.SUM)
        stats[0] = tmax[0]
    dt_max, rays_all, cam_all = [float(v) for v in stats.cpu()]

    if rank == 0:
        # dominant kernel: k_ptrace<false,false>, the closest-hit persistent tracer, launched for the path segments
        # ("trace_closest") and - unless the library can answer them with any-hit rays ("nee_mis_any": scenes without emitting
        # primitives and media, like this one) - for the MIS rays of direct lighting ("nee_mis")
        def closest(p):
            return p["trace_closest"]["launches"] + p["nee_mis"]["launches"], p["trace_closest"]["total_ms"] + p["nee_mis"]["total_ms"]
        mis_closest = 1 if profx["nee_mis"]["launches"] > 0 else 0
        traced = max(1, cc["total_paths"] + mis_closest * cc["mis_rays"])  # counting pass: every one of them went through the counting variant
        nodes_per_ray = cc["nodes_visited"] / traced
        items_per_ray = cc["prims_tested"] / traced
        bytes_per_ray = 64.0 + 32.0 * nodes_per_ray + 48.0 * items_per_ray
        launches, total_ms = closest(profx)
        launches = max(1, launches)
        avg_ms = total_ms / launches
        rays_per_launch = (cx["total_paths"] + mis_closest * cx["mis_rays"]) / launches  # incl. the ~3 % of segments the finish kernel traces itself
        achieved = bytes_per_ray * rays_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        t_launches, t_ms = closest(prof)
        t_avg_ms = t_ms / max(1, t_launches)
        t_rays_per_launch = (c["total_paths"] + mis_closest * c["mis_rays"]) / max(1, t_launches)
        t_achieved = bytes_per_ray * t_rays_per_launch / (t_avg_ms * 1e-3) / 1e9 if t_avg_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic("k_ptrace<false, false>")
        out = {
            "metric": "Mrays/s (incoherent bounces)",
            "value": None if args.roofline_only else rays_all / dt_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: %d-triangle torus-knot mesh (dragon-class stand-in), %s, light-map sky, %dx%d, "
                                   "%d spp per step per GPU, integrator %d (Li_VolPath=3)" %
                                   (2 * args.segments * args.segments, ["rough glass (microfacet)", "plastic", "specular glass", "matte"][args.variant],
                                    W, H, args.spp, args.integrator),
                       "triangles": 2 * args.segments * args.segments, "width": W, "height": H,
                       "bvh_nodes": bvh["nodes"], "bvh_items": bvh["items"], "bvh_bytes": bvh["bytes"],
                       "scene_upload_s": round(t_up, 3), "parallelism": "samples round-robin over %d GPU(s), scene replicated" % world},
            "samples_per_s": cam_all / dt_max,
            "rays_per_sample": rays_all / max(1.0, cam_all),
            "roofline": {"bound": "hbm", "kernel": "k_ptrace<false,false> (persistent tracer, closest hit)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per_ray * rays_per_launch,
                         "bytes_per_ray": bytes_per_ray, "nodes_per_ray": nodes_per_ray, "items_per_ray": items_per_ray,
                         "avg_launch_ms": avg_ms, "rays_per_launch": rays_per_launch, "launches": launches,
                         "measured": "HIP events on the launch stream, passes run one at a time after the timed region (as rocprofv3 serialises them)",
                         "timed_region": {"avg_launch_ms": t_avg_ms, "launches": t_launches, "achieved": t_achieved, "frac": t_achieved / HBM_PEAK_GBS,
                                          "note": "six passes overlap on six streams: a launch shares the GPU with other passes' kernels"}},
            "kernels_ms": {k: round(v["total_ms"], 3) for k, v in prof.items() if v["launches"]},
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, W, H, args.integrator, args.seed)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    r.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
