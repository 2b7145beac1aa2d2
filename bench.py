#!/usr/bin/env python3
"""Headline benchmark of the per-ray hot path (BASELINE.json: Mrays/s, incoherent bounces).

--config picks one of BASELINE.json's configs[] (default 1, the one the metric is quoted on):
  0  Cornell box 512x512 (the reference's CPU-runnable plumbing case)
  1  871,200-triangle torus-knot mesh ("dragon-class" stand-in: the author's assets are not shipped), rough-glass
     microfacet material, light-map sky, 1920x1080
  2  indoor room: 40 rectangle / SDF-box primitives, 8 x 100,352-triangle meshes, 64 emitters, 1920x1080
  3  SPH dam-break stand-in: 4,000,000 particle spheres, 3840x2160
  4  subsurface-scattering knot (299,538 triangles) + three BSSRDF spheres, 1920x1080
  5  the volumetric-caustics scene (second half of configs[4]), 1920x1080
All with Li_VolPath, the reference's shipped integrator (main.cpp:447).  One *step* is one batch of --spp samples of every
pixel through the wavefront pipeline (raygen -> {trace, shade, direct light} per bounce -> film); scene, BVH and film are
resident in HBM before the timed region starts.

value = incoherent rays per second: every scene query of depth >= 1 (path segments after the camera ray, shadow rays,
MIS rays, BSSRDF probe rays) of all ranks / wall time.  The coherent camera rays are reported beside it
(mrays_primary), as are the closest-hit and any-hit shares of the incoherent rays.

N > 1 (one rank per GPU, scene replicated, no exchange while rendering, the film assembled on rank 0 inside the timed
region - a gather of packed tiles or a sum reduce, pathtracer_amd.dist.FilmGather):
  --scaling weak    (default) every rank renders --spp samples of every pixel: sample s -> rank s mod N
  --scaling strong  --spp samples per step in TOTAL, split over the ranks by --split tiles (8x8-pixel tiles dealt round-robin,
                    the reduced film is bit-identical to the 1-GPU film; default) or --split samples

    python bench.py --gpus 1 --steps 8 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
ROOFLINE_PASSES = 2  # passes of the roofline leg, each alone on the GPU (kernels_ms_exclusive_per_step divides by this)

# config -> (scene family, p0, p1, width, height, samples per step, description)
CONFIGS = {
    0: ("cornell", 0, 0, 512, 512, 64, "configs[0]: Cornell box (6 rectangles, 2 SDF boxes, disk emitter), diffuse + emissive"),
    1: ("knot", 660, 0, 1920, 1080, 32, "configs[1]: 871,200-triangle torus-knot mesh (dragon-class stand-in), rough glass (microfacet), light-map sky"),
    2: ("room", 224, 0, 1920, 1080, 32, "configs[2]: indoor room, 40 rectangle / SDF-box primitives + 8 x 100,352-triangle meshes + 64 emitters"),
    3: ("fluid", 4000000, 0, 3840, 2160, 8, "configs[3]: SPH dam-break stand-in, 4,000,000 particle spheres in 6 clouds, light-map sky"),
    4: ("subsurface", 387, 0, 1920, 1080, 16, "configs[4]: subsurface-scattering knot (299,538 triangles) + 3 BSSRDF spheres, light-map sky"),
    5: ("volcaustic", 0, 0, 1920, 1080, 32, "configs[4], second half: VolumetricCausticsScene (glass sphere in a homogeneous medium, 2 sphere emitters)"),
}


def newest_summary():
    """The committed PMC summary that was collected from THIS tree's kernel sources (`_csrc_hash`), the last by name if there are
    several; else the last by name (measured_traffic then reports null: stale counters are not quoted)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return None, None
    loaded = []
    for f in files:
        try:
            loaded.append((json.load(open(f)), os.path.relpath(f, ROOT)))
        except Exception:
            pass
    if not loaded:
        return None, None
    try:
        from pathtracer_amd.capi import kernel_source_hash
        h = kernel_source_hash()
        fresh = [x for x in loaded if x[0].get("_csrc_hash") == h]
        if fresh:
            return fresh[-1]
    except Exception:
        pass
    return loaded[-1]


def measured_traffic(summary, kernel, config):
    """HBM-side bytes per launch of `kernel` from the newest committed PMC summary (profiles/r*_summary.json, written by
    tools/gpu_profile.sh -> tools/prof_summary.py --json from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
    this same command).  rocprofv3 reports KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, hence the factor 2
    (MI355X_MICROARCH.md, HBM section).  None if the summary is of another config, or was collected from other kernel
    sources than this tree's (`_csrc_hash`): a stale figure is not reported."""
    if not summary or summary.get("_config", 1) != config:
        return None
    from pathtracer_amd.capi import kernel_source_hash
    if summary.get("_csrc_hash") != kernel_source_hash():
        return None  # counters of other kernel sources than this tree's: stale, not reported
    try:
        if kernel not in summary:  # a prefix: the tracer's kernels carry further template arguments (fast top level, waves per SIMD)
            names = [n for n in summary if n.startswith(kernel) and isinstance(summary[n], dict) and summary[n].get("counters")]
            kernel = max(names, key=lambda n: summary[n].get("dispatches") or summary[n].get("calls") or 0)
        k = summary[kernel]
        c = k["counters"]
        n = max(1, int(k.get("dispatches") or k.get("calls") or 1))
        return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / n
    except Exception:
        return None


def classify(c, mis_any, shadow_closest=False):
    """ray classes from the counters: coherent camera rays; incoherent closest-hit and any-hit scene queries"""
    primary = c["camera_rays"]
    closest = c["total_paths"] - c["camera_rays"] + c["probe_rays"] + (0 if mis_any else c["mis_rays"]) + (c["shadow_rays"] if shadow_closest else 0)
    anyhit = (0 if shadow_closest else c["shadow_rays"]) + (c["mis_rays"] if mis_any else 0)
    return primary, closest, anyhit


def cpu_baseline(scene, width, height, integrator, seed):
    """The oracle (CPU restatement of the reference path: kind "port", not a host build of the reference, which needs
    the CUDA toolkit) on this box's host cores, bounded sample."""
    from oracle import orc

    cores = max(1, min(16, os.cpu_count() or 1))
    osc = orc.OracleScene(scene.flat, scene.rand_calls)
    # a reduced-resolution frame of the SAME scene/camera/integrator, repeated until ~12 s of CPU work
    w, h = max(16, width // 4), max(16, height // 4)
    dt, spp = 0.0, 0
    while dt < 12.0 and spp < 512:
        ns = 1 if spp == 0 else int(min(64, max(1, (12.5 - dt) / max(dt / spp, 1e-3))))
        t0 = time.time()
        osc.render(scene.camera, w, h, seed, spp, ns, integrator, cores)
        dt += time.time() - t0
        spp += ns
    c = osc.counters()
    osc.close()
    rays_all = c["total_paths"] + c["shadow_rays"] + c["mis_rays"] + c["probe_rays"]
    return {"value": (rays_all - c["camera_rays"]) / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "mrays_all": rays_all / dt / 1e6,
            "sample": "%dx%d x %d spp of the same scene/camera/integrator, %.1f s, %d rays (%d of them camera rays)" % (w, h, spp, dt, rays_all, c["camera_rays"])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="index into BASELINE.json configs[] (5 = second scene of configs[4])")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per step (weak: per GPU; strong: in total); 0 = the config's default")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--split", choices=["tiles", "samples"], default=None, help="how N > 1 ranks share a step (strong: tiles by default; weak is always samples)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--size", type=int, default=0, help="scene size parameter (config 1: knot tessellation, 2 * s * s triangles)")
    ap.add_argument("--variant", type=int, default=-1, help="config 1: 0 rough glass, 1 plastic, 2 specular glass, 3 matte")
    ap.add_argument("--integrator", type=int, default=3)
    ap.add_argument("--bvh", choices=["host", "device"], default="host", help="geometry BVH builder: host binned SAH, or the device LBVH (SURVEY 8f row 3)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on one GPU: take the N > 1 code path (RCCL process group, barriers, all-reduce of the film) with a world of one rank")
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
    if world > 1 and os.environ.get("PTR_BENCH_REHEARSE_ON_GPU0"):
        # rehearsal of N ranks on a one-GPU box: every rank renders on GPU 0, the collectives go over gloo (RCCL wants a GPU per rank)
        local_rank = 0
        torch.cuda.set_device(0)
        dist = pdist.init_process_group("gloo")
    elif world > 1 or args.force_dist:
        dist = pdist.init_process_group("nccl")
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    fam, p0, p1, W, H, spp, label = CONFIGS[args.config]
    W, H = args.width or W, args.height or H
    p0 = args.size or p0
    p1 = args.variant if args.variant >= 0 else p1
    spp = args.spp or spp
    split = "samples" if args.scaling == "weak" else (args.split or "tiles")
    scene = pa.BuiltScene(fam, W, H, p0, p1)
    fs = scene.flat.contents
    n_tris = sum(fs.meshes[i].n_triangles for i in range(fs.n_meshes))
    n_parts = sum(fs.clouds[i].n_particles for i in range(fs.n_clouds))
    # Li_VolPath's shadow / MIS rays walk through primitives without a material (emitters, medium boundaries) inside
    # the shade stage's own kernels; otherwise they go through the persistent tracer
    prims = [fs.primitives[i] for i in range(fs.n_primitives)]
    bare = any(p.no_mat and not p.mat.is_emissive for p in prims)          # a medium boundary without a material: rays pass through it
    emitters = any((not p.no_mat) and p.mat.is_emissive for p in prims)
    nee_closest_ok = os.environ.get("PTR_NEE_CLOSEST", "1") != "0"
    walks = args.integrator == 3 and (bare or (emitters and not nee_closest_ok))
    # Li_VolPath, emitters but no bare boundary: a shadow ray is ONE closest-hit query of the persistent tracer (dev_nee.h)
    shadow_closest = args.integrator == 3 and emitters and not bare and nee_closest_ok
    r = pa.Renderer(local_rank)
    pa.set_bvh_build_mode(args.bvh)
    t_up = time.time()
    r.upload(scene)
    t_up = time.time() - t_up
    bvh_build = r.bvh_build_info()
    r.set_camera(scene.camera)

    def new_film():
        r.film_init(W, H, args.seed)
        if split == "tiles":
            r.tiles(rank, world)
        else:
            r.shard(rank, world)

    new_film()
    bvh = r.bvh_info()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        # weak: spp samples of every pixel on every rank (this rank takes the sample indices = rank mod world of world * spp);
        # strong: spp samples in total (this rank's tiles, or its share of the sample indices)
        r.render(world * spp if args.scaling == "weak" else spp, args.integrator)

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
    sc_count = r.pipeline_stats()
    r.profile_enable(timing=True, count_traversal=False)
    r.profile_reset()
    r.counters_reset()
    new_film()  # the timed region starts from a clean film

    we_t, acc_t, smp_t = pdist.film_tensors(r, dev)
    # the whole job's film is assembled on rank 0 in tensors of its own (tiles: one gather of packed tiles; samples: one sum
    # reduce); the ranks' own films stay as they are
    gather = pdist.FilmGather(dist, W, H, split, dev, renderer=r) if dist is not None else None

    barrier()
    t0 = time.perf_counter()
    for _ in range(0 if args.roofline_only else args.steps):
        step()
    r.sync()
    if dist is not None:
        total = args.steps * (world * spp if args.scaling == "weak" else spp)
        gather.run(we_t, acc_t, smp_t, total_samples=total)
    barrier()
    dt = time.perf_counter() - t0

    c = r.counters()
    prof = r.profile()
    mis_any = prof["nee_mis_any"]["launches"] > 0 or (not walks and prof["nee_mis"]["launches"] == 0)
    primary, sec_closest, sec_any = classify(c, mis_any, shadow_closest)

    # roofline leg (untimed, rank 0's GPU only needs it but every rank runs it to stay in step): passes rendered one
    # at a time, so each launch has the GPU to itself - the condition rocprofv3 --kernel-trace measures under (it
    # serialises dispatches), whereas in the timed region six passes overlap and every launch shares the GPU.
    r.profile_reset()
    r.counters_reset()
    for _ in range(ROOFLINE_PASSES):
        step()
        r.sync()
    cx = r.counters()
    sx = r.pipeline_stats()
    profx = r.profile()

    stats = torch.tensor([dt, float(primary), float(sec_closest), float(sec_any)], dtype=torch.float64, device=dev)
    if dist is not None:
        tmax = stats[0:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats[1:], op=dist.ReduceOp.SUM)
        stats[0] = tmax[0]
    dt_max, prim_all, clo_all, any_all = [float(v) for v in stats.cpu()]

    if rank == 0:
        summary, summary_src = newest_summary()
        # ---- dominant kernel: k_ptrace<false,false>, the closest-hit persistent tracer, launched for the path segments
        # ("trace_closest") and - when they need a closest hit and do not walk - for the MIS rays of direct lighting ("nee_mis")
        mis_closest = 0 if (mis_any or walks) else 1

        sh_closest = 1 if shadow_closest else 0

        def closest(p):
            return (p["trace_closest"]["launches"] + mis_closest * p["nee_mis"]["launches"] + sh_closest * p["nee_shadow"]["launches"],
                    p["trace_closest"]["total_ms"] + mis_closest * p["nee_mis"]["total_ms"] + sh_closest * p["nee_shadow"]["total_ms"])
        traced = max(1, cc["total_paths"] + mis_closest * cc["mis_rays"] + sh_closest * cc["shadow_rays"])  # counting pass: every bounce is staged, all of them go through the counting variant
        nodes_per_ray = cc["nodes_visited"] / traced
        items_per_ray = cc["prims_tested"] / traced
        bytes_per_ray = 64.0 + 32.0 * nodes_per_ray + 48.0 * items_per_ray
        launches, total_ms = closest(profx)
        launches = max(1, launches)
        avg_ms = total_ms / launches
        # rays the tracer launches carried: the segments the finish kernel traces inside itself are not theirs
        rays_per_launch = (cx["total_paths"] - sx["finish_paths"] + mis_closest * cx["mis_rays"] + sh_closest * cx["shadow_rays"]) / launches
        achieved = bytes_per_ray * rays_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        t_launches, t_ms = closest(prof)
        t_avg_ms = t_ms / max(1, t_launches)
        traffic = measured_traffic(summary, "k_ptrace<false, false, false", args.config)
        # ---- the any-hit tracer k_ptrace<true,false> (shadow rays; MIS rays where any hit answers them): the largest kernel by
        # exclusive time on configs[1].  Same pricing, B = 52 (48 B ray in, 4 B answer out) + 32 per child box + 48 per item, with
        # its own per-ray counts from the counting pass (an any-hit ray ends at its first hit).
        any_launches = (0 if shadow_closest or walks else profx["nee_shadow"]["launches"]) + profx["nee_mis_any"]["launches"]
        any_ms = (0.0 if shadow_closest or walks else profx["nee_shadow"]["total_ms"]) + profx["nee_mis_any"]["total_ms"]
        roofline_any = None
        if any_launches and sc_count["any_rays"]:
            a_nodes = sc_count["any_nodes"] / sc_count["any_rays"]
            a_items = sc_count["any_items"] / sc_count["any_rays"]
            a_bytes = 52.0 + 32.0 * a_nodes + 48.0 * a_items
            kinds_any = (0 if shadow_closest or walks else 1) + (1 if mis_any else 0)  # of the two direct-light ray kinds, how many are any-hit rays
            a_rays = ((0 if shadow_closest or walks else cx["shadow_rays"]) + (cx["mis_rays"] if mis_any else 0)
                      - sx["finish_nee_rays"] * kinds_any / 2.0) / any_launches  # the finish kernel traces its direct-light rays itself
            a_avg = any_ms / any_launches
            a_ach = a_bytes * a_rays / (a_avg * 1e-3) / 1e9 if a_avg > 0 else 0.0
            roofline_any = {"bound": "hbm", "kernel": "k_ptrace<true,false> (persistent tracer, any hit: shadow rays + MIS rays that any hit answers)",
                            "achieved": a_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a_ach / HBM_PEAK_GBS,
                            "traffic": measured_traffic(summary, "k_ptrace<true, false, false", args.config), "traffic_unit": "bytes per launch",
                            "bytes_per_ray": a_bytes, "nodes_per_ray": a_nodes, "items_per_ray": a_items, "avg_launch_ms": a_avg,
                            "rays_per_launch": a_rays, "launches": any_launches,
                            "note": "the direct-light rays the finish kernel traces inside itself are taken off (half of them per any-hit ray kind)"}
        # ---- shade stage (k_shade + k_route): SURVEY 8(d) B_shade = 192 B per shaded vertex, + the 192-B direct-light record
        # of the vertices that write one
        sh_launches = max(1, profx["shade"]["launches"])
        sh_avg_ms = profx["shade"]["total_ms"] / sh_launches
        verts_per_launch = sx["shaded_vertices"] / sh_launches
        bytes_per_vertex = 192.0 + 192.0 * sx["nee_records"] / max(1, sx["shaded_vertices"])
        sh_achieved = bytes_per_vertex * verts_per_launch / (sh_avg_ms * 1e-3) / 1e9 if sh_avg_ms > 0 else 0.0
        sh_traffic = measured_traffic(summary, "k_shade<0", args.config) or measured_traffic(summary, "k_shade<false>", args.config)  # (k_shade<0, 3>: mode, integrator)
        cam_rate = prim_all / dt_max if not args.roofline_only else None
        out = {
            "metric": "Mrays/s (incoherent bounces)",
            "value": None if args.roofline_only else (clo_all + any_all) / dt_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s, %dx%d, %d spp per step %s, integrator %d (Li_VolPath=3)" %
                                   (label, W, H, spp, "per GPU" if args.scaling == "weak" else "in total", args.integrator),
                       "config_index": args.config, "triangles": n_tris, "particles": n_parts, "primitives": fs.n_primitives,
                       "width": W, "height": H, "bvh_nodes": bvh["nodes"], "bvh_items": bvh["items"], "bvh_bytes": bvh["bytes"],
                       "scene_upload_s": round(t_up, 3), "bvh_builder": bvh_build["builder"], "bvh_build_ms": round(bvh_build["build_ms"], 3),
                       "bvh_depth": bvh_build["depth"],
                       "parallelism": "%s over %d GPU(s), scene replicated, %s" %
                                      ("8x8-pixel tiles round-robin" if split == "tiles" else "samples round-robin", world,
                                       "one gather of packed tiles to rank 0" if split == "tiles" else "one sum reduce of the accumulators to rank 0")},
            "value_definition": "scene queries of depth >= 1 (path segments after the camera ray + shadow + MIS + BSSRDF probe rays) per second; camera rays excluded "
                                "(since round 2; round 1's `value` counted camera rays too: that figure is `mrays_all`, kept under that key in every round since)",
            "mrays_primary": None if args.roofline_only else prim_all / dt_max / 1e6,
            "mrays_secondary_closest": None if args.roofline_only else clo_all / dt_max / 1e6,
            "mrays_secondary_any": None if args.roofline_only else any_all / dt_max / 1e6,
            "mrays_all": None if args.roofline_only else (prim_all + clo_all + any_all) / dt_max / 1e6,
            "samples_per_s": cam_rate,
            "rays_per_sample": (prim_all + clo_all + any_all) / max(1.0, prim_all),
            "roofline": {"bound": "hbm", "kernel": "k_ptrace<false,false> (persistent tracer, closest hit)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": summary_src if traffic is not None else None,
                         "algorithmic_bytes_per_launch": bytes_per_ray * rays_per_launch,
                         "bytes_per_ray": bytes_per_ray, "nodes_per_ray": nodes_per_ray, "items_per_ray": items_per_ray,
                         "avg_launch_ms": avg_ms, "rays_per_launch": rays_per_launch, "launches": launches,
                         "measured": "HIP events on the launch stream, passes run one at a time after the timed region (as rocprofv3 serialises them); "
                                     "rays the finish kernel traces inside itself are not counted",
                         "timed_region": {"avg_launch_ms": t_avg_ms, "launches": t_launches,
                                          "note": "six passes overlap on six streams: a launch shares the GPU with other passes' kernels"}},
            "roofline_any": roofline_any,
            "roofline_shade": {"bound": "hbm", "kernel": "k_shade + k_route (one integrator iteration per path vertex, queue routing)",
                               "achieved": sh_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sh_achieved / HBM_PEAK_GBS,
                               "traffic": sh_traffic, "traffic_unit": "bytes per launch (k_shade alone)",
                               "bytes_per_vertex": bytes_per_vertex, "vertices_per_launch": verts_per_launch, "avg_launch_ms": sh_avg_ms,
                               "launches": sh_launches, "nee_record_fraction": sx["nee_records"] / max(1, sx["shaded_vertices"]),
                               "note": "VALU-issue bound, not bandwidth bound (DESIGN.md 4.2): the fraction prices its traffic, the time is instructions"},
            "kernels_ms": {k: round(v["total_ms"], 3) for k, v in prof.items() if v["launches"]},
            "kernels_ms_exclusive_per_step": {k: round(v["total_ms"] / ROOFLINE_PASSES, 3) for k, v in profx.items() if v["launches"]},
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
