"""ctypes view of include/ptr_api.h (the C-ABI of lib/libptr_hip.so) and of the host-side
scene mirror lib/libpt_host.so.

There is no CPU fallback: if the HIP library is missing or does not load, importing a symbol
through :func:`hip` raises, loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
HIP_LIB = os.environ.get("PTR_HIP_LIB") or os.path.join(LIB_DIR, "libptr_hip.so")  # override: kernel-variant experiments
HOST_LIB = os.path.join(LIB_DIR, "libpt_host.so")

PTR_OK, PTR_ERR_INVALID, PTR_ERR_HIP, PTR_ERR_STATE, PTR_ERR_LIMIT = 0, 1, 2, 3, 4
LI_DIRECT, LI_PATH, LI_PATHSAMPLED, LI_VOLPATH = 0, 1, 2, 3
TONEMAP_REINHARD, TONEMAP_EXPONENTIAL, TONEMAP_NAUGHTYDOG = 0, 1, 2
SHAPE_SPHERE, SHAPE_RECTANGLE, SHAPE_DISK, SHAPE_BOX, SHAPE_MESH, SHAPE_PARTICLECLOUD = 0, 1, 2, 3, 4, 5
SHAPE_SPHERE_PROCEDURAL, SHAPE_BOX_PROCEDURAL, SHAPE_COMPONENT_PROCEDURAL = 6, 7, 8
MAT_NONE, MAT_MATTE, MAT_MIRROR, MAT_GLASS, MAT_METAL, MAT_TRANSLUCENT, MAT_PLASTIC, MAT_UBER = range(8)
MAT_SUBSURFACE, MAT_KDSUBSURFACE, MAT_PLAYGROUND, MAT_DISNEY = 8, 9, 10, 11
PROF_KERNELS = 12

f32, i32, u64, i64 = C.c_float, C.c_int32, C.c_uint64, C.c_int64
fptr, iptr = C.POINTER(C.c_float), C.POINTER(C.c_int32)


class PtrTransform(C.Structure):
    _fields_ = [("m", f32 * 16), ("mInv", f32 * 16)]


class PtrTexture(C.Structure):
    _fields_ = [("type", i32), ("spec", f32 * 3), ("fval", f32), ("image", i32)]


class PtrMaterial(C.Structure):
    _fields_ = [("type", i32), ("is_emissive", i32), ("n_textures", i32), ("id", i32),
                ("textures", PtrTexture * 16), ("vals", f32 * 16)]


class PtrMedium(C.Structure):
    _fields_ = [("sigma_a", f32 * 3), ("sigma_s", f32 * 3), ("g", f32), ("is_valid", i32)]


class PtrPrimitive(C.Structure):
    _fields_ = [("shape_type", i32), ("reverse_orientation", i32), ("geom", i32), ("no_mat", i32),
                ("to_world", PtrTransform), ("params", f32 * 8), ("mat", PtrMaterial), ("medium", PtrMedium)]


class PtrMesh(C.Structure):
    _fields_ = [("n_triangles", i32), ("n_vertices", i32), ("n_uvs", i32), ("_pad", i32),
                ("p", fptr), ("indices", iptr), ("uv", fptr), ("to_world", PtrTransform)]


class PtrCloud(C.Structure):
    _fields_ = [("n_particles", i32), ("radius", f32), ("positions", fptr)]


class PtrImage(C.Structure):
    _fields_ = [("width", i32), ("height", i32), ("rgb", fptr)]


class PtrEnvMap(C.Structure):
    _fields_ = [("light_to_world", PtrTransform), ("width", i32), ("height", i32), ("nu", i32), ("nv", i32),
                ("texels", fptr), ("func", fptr), ("cdf", fptr), ("func_int", fptr), ("marg_cdf", fptr),
                ("marg_int", f32), ("_pad", i32)]


class PtrDistantLight(C.Structure):
    _fields_ = [("L", f32 * 3), ("w", f32 * 3)]


class PtrFlatScene(C.Structure):
    _fields_ = [("n_primitives", i32), ("n_meshes", i32), ("n_clouds", i32), ("n_images", i32),
                ("primitives", C.POINTER(PtrPrimitive)), ("meshes", C.POINTER(PtrMesh)),
                ("clouds", C.POINTER(PtrCloud)), ("images", C.POINTER(PtrImage)),
                ("camera_medium", PtrMedium), ("has_envmap", i32), ("_pad", i32), ("envmap", PtrEnvMap),
                ("n_distant", i32), ("_pad2", i32), ("distant", C.POINTER(PtrDistantLight))]


class PtrCamera(C.Structure):
    _fields_ = [("position", f32 * 3), ("lower_left", f32 * 3), ("horizontal", f32 * 3), ("vertical", f32 * 3),
                ("lens_radius", f32), ("u", f32 * 3), ("v", f32 * 3), ("w", f32 * 3)]


COUNTER_NAMES = ["total_paths", "misses", "surface_hits", "medium_hits", "light_hits", "zero_radiance",
                 "shadow_rays", "mis_rays", "probe_rays", "camera_rays", "nodes_visited", "prims_tested"]


class PtrCounters(C.Structure):
    _fields_ = [(n, u64) for n in COUNTER_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_NAMES}


class PtrPipelineStats(C.Structure):
    _fields_ = [(n, u64) for n in ("finish_paths", "finish_nee_rays", "shaded_vertices", "nee_records", "any_nodes", "any_items", "any_rays", "fast_top_deferred")]


class PtrBvhBuildInfo(C.Structure):
    _fields_ = [("builder", C.c_int32), ("depth", C.c_int32), ("build_ms", C.c_double), ("sah_boxes", C.c_double), ("sah_items", C.c_double),
                ("sah_leaves", C.c_double)]


class PtrKernelProfile(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", u64), ("total_ms", C.c_double), ("items", u64)]


# every entry point include/ptr_api.h declares: name -> (restype, argtypes)
CTX = C.c_void_p
HIP_SIGNATURES = {
    "ptr_create": (C.c_int, [C.c_int, C.POINTER(CTX)]),
    "ptr_destroy": (C.c_int, [CTX]),
    "ptr_last_error": (C.c_char_p, []),
    "ptr_set_stream": (C.c_int, [CTX, C.c_void_p]),
    "ptr_scene_upload": (C.c_int, [CTX, C.POINTER(PtrFlatScene)]),
    "ptr_envmap_build": (C.c_int, [C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_void_p)]),
    "ptr_envmap_view": (C.c_int, [C.c_void_p, C.POINTER(PtrEnvMap), C.POINTER(C.c_double)]),
    "ptr_envmap_free": (C.c_int, [C.c_void_p]),
    "ptr_scene_build_mode": (C.c_int, [C.c_int]),
    "ptr_bvh_build_info": (C.c_int, [CTX, C.POINTER(PtrBvhBuildInfo)]),
    "ptr_scene_built_info": (C.c_int, [C.c_void_p, C.POINTER(PtrBvhBuildInfo)]),
    "ptr_scene_build": (C.c_int, [C.POINTER(PtrFlatScene), C.POINTER(C.c_void_p)]),
    "ptr_scene_built_hash": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "ptr_scene_upload_built": (C.c_int, [CTX, C.c_void_p]),
    "ptr_scene_built_free": (C.c_int, [C.c_void_p]),
    "ptr_camera_set": (C.c_int, [CTX, C.POINTER(PtrCamera)]),
    "ptr_film_init": (C.c_int, [CTX, C.c_int, C.c_int, u64]),
    "ptr_film_shard": (C.c_int, [CTX, C.c_int, C.c_int]),
    "ptr_film_tiles": (C.c_int, [CTX, C.c_int, C.c_int]),
    "ptr_render": (C.c_int, [CTX, C.c_int, C.c_int]),
    "ptr_sync": (C.c_int, [CTX]),
    "ptr_film_read": (C.c_int, [CTX, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptr_film_rgb8": (C.c_int, [CTX, C.c_float, C.c_int, C.c_void_p]),
    "ptr_film_write": (C.c_int, [CTX, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "ptr_film_device_ptrs": (C.c_int, [CTX, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "ptr_film_tiles_packed_pixels": (C.c_int, [CTX, C.c_int, C.POINTER(C.c_int64)]),
    "ptr_film_tiles_pack": (C.c_int, [CTX, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptr_film_tiles_unpack": (C.c_int, [CTX, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptr_film_fill_samples": (C.c_int, [CTX, C.c_void_p, C.c_int64]),
    "ptr_counters_read": (C.c_int, [CTX, C.POINTER(PtrCounters)]),
    "ptr_counters_reset": (C.c_int, [CTX]),
    "ptr_generate_rays": (C.c_int, [CTX, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ptr_trace_closest": (C.c_int, [CTX, C.c_int] + [C.c_void_p] * 8),
    "ptr_trace_closest_noalloc": (C.c_int, [CTX, C.c_int] + [C.c_void_p] * 8),
    "ptr_trace_occluded": (C.c_int, [CTX, C.c_int] + [C.c_void_p] * 4),
    "ptr_pipeline_stats_read": (C.c_int, [CTX, C.POINTER(PtrPipelineStats)]),
    "ptr_unit_eval": (C.c_int, [CTX, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "ptr_unit_bsdf": (C.c_int, [CTX, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7),
    "ptr_profile_enable": (C.c_int, [CTX, C.c_int]),
    "ptr_profile_reset": (C.c_int, [CTX]),
    "ptr_profile_read": (C.c_int, [CTX, C.POINTER(PtrKernelProfile), C.c_int, C.POINTER(C.c_int)]),
    "ptr_bvh_info": (C.c_int, [CTX, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
}
HOST_SIGNATURES = {
    "pth_scene_build": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pth_lightmap_scene": (C.c_void_p, [C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]),
    "SetLightMapBuilderC": (None, [C.c_int]),
    "pth_scene_flat": (C.POINTER(PtrFlatScene), [C.c_void_p]),
    "pth_scene_camera": (None, [C.c_void_p, C.POINTER(PtrCamera)]),
    "pth_scene_rand_calls": (C.c_int, [C.c_void_p]),
    "pth_scene_free": (None, [C.c_void_p]),
    "pth_transform": (None, [C.c_char_p, f32, f32, f32, f32, C.POINTER(PtrTransform)]),
    "pth_png_encode": (C.c_long, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long]),
    "pth_image_write": (C.c_int, [CTX, C.c_int, C.c_int, C.c_char_p, C.c_float, C.c_int]),
    "pth_ppm_read": (C.c_int, [C.c_char_p, C.c_void_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pth_parse_double": (C.c_int, [C.c_char_p, C.POINTER(C.c_float)]),
    "pth_obj_load": (C.c_void_p, [C.c_char_p, C.c_int]),
    "pth_obj_count": (C.c_int, [C.c_void_p]),
    "pth_obj_mesh": (None, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(fptr), C.POINTER(fptr), C.POINTER(fptr), C.POINTER(iptr)]),
    "pth_obj_mtl": (C.c_long, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_long]),
    "pth_obj_free": (None, [C.c_void_p]),
    "pth_mtl_load": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_int]),
    "pth_mtl_count": (C.c_int, [C.c_void_p]),
    "pth_mtl_name": (C.c_long, [C.c_void_p, C.c_int, C.c_char_p, C.c_long]),
    "pth_mtl_find": (C.c_int, [C.c_void_p, C.c_char_p]),
    "pth_mtl_value": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_long]),
    "pth_mtl_spectrum": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, fptr]),
    "pth_mtl_material": (None, [C.c_void_p, C.c_int, C.POINTER(C.c_int), fptr]),
    "pth_mtl_free": (None, [C.c_void_p]),
    "pth_image_load": (C.c_void_p, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(fptr)]),
    "pth_image_free": (None, [C.c_void_p]),
    "pth_jpeg_idct": (None, [C.POINTER(C.c_short), C.POINTER(C.c_ubyte)]),
    "pth_exr_read": (C.c_void_p, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pth_exr_free": (None, [C.c_void_p]),
    "pth_make_sky": (None, [C.c_int, C.c_int, C.c_void_p]),
    "pth_scene_sky_exr": (None, [C.c_char_p]),
    "pth_particles_load": (C.c_void_p, [C.c_char_p, C.c_char_p]),
    "pth_particles_get": (fptr, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "pth_particles_free": (None, [C.c_void_p]),
    "pth_camera_config": (None, [fptr, fptr, fptr, f32, f32, f32, C.POINTER(PtrCamera)]),
}

_hip = None
_host = None


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


def hip():
    """The C-ABI library.  Raises RuntimeError if it has not been built (no fallback exists)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError(
                "pathtracer_amd: %s is missing - build it with `make -C pathtracer_amd` "
                "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % HIP_LIB)
        _hip = _bind(C.CDLL(HIP_LIB, mode=C.RTLD_GLOBAL), HIP_SIGNATURES)
    return _hip


def host():
    """Host-side mirror of the reference's scene API (links against the C-ABI library)."""
    global _host
    if _host is None:
        hip()
        if not os.path.exists(HOST_LIB):
            raise RuntimeError("pathtracer_amd: %s is missing - build it with `make -C pathtracer_amd`" % HOST_LIB)
        _host = _bind(C.CDLL(HOST_LIB), HOST_SIGNATURES)
    return _host


MULTIGPU_LIB = os.path.join(LIB_DIR, "libpt_multigpu.so")
MULTIGPU_SIGNATURES = {
    "RenderGroupCreate": (C.c_void_p, [C.c_int, C.POINTER(C.c_int), C.c_int]),
    "RenderGroupFree": (None, [C.c_void_p]),
    "RenderGroupSize": (C.c_int, [C.c_void_p]),
    "RenderGroupContext": (CTX, [C.c_void_p, C.c_int]),
    "RenderGroupUpload": (C.c_int, [C.c_void_p, C.POINTER(PtrFlatScene)]),
    "RenderGroupCamera": (C.c_int, [C.c_void_p, C.POINTER(PtrCamera)]),
    "RenderGroupFilmInit": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint64]),
    "RenderGroupRenderSamples": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "RenderGroupReduce": (C.c_int, [C.c_void_p]),
    "RenderGroupSync": (C.c_int, [C.c_void_p]),
    "RenderGroupFilmRead": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "RenderGroupLastError": (C.c_char_p, []),
}
_multigpu = None


def multigpu():
    """The C++ multi-GPU host driver (host/pt_multigpu.cpp): RCCL called directly, one process for all GPUs of the node."""
    global _multigpu
    if _multigpu is None:
        host()
        if not os.path.exists(MULTIGPU_LIB):
            raise RuntimeError("pathtracer_amd: %s is missing - build it with `make -C pathtracer_amd`" % MULTIGPU_LIB)
        _multigpu = _bind(C.CDLL(MULTIGPU_LIB), MULTIGPU_SIGNATURES)
    return _multigpu


def bvh_build_info_dict(o):
    return {"builder": "device_lbvh" if o.builder == 1 else "host_sah", "build_ms": o.build_ms, "depth": o.depth, "sah_boxes": o.sah_boxes,
            "sah_items": o.sah_items, "sah_leaves": o.sah_leaves}


class PtrError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "ptr error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc != PTR_OK:
        raise PtrError(rc, hip().ptr_last_error().decode("utf-8", "replace"))


def kernel_source_hash():
    """sha256 over the kernel sources (csrc/*, sorted by name) and the Makefile (compile flags) the built library comes from: profiles/*_summary.json records
    it when counters are collected, and bench.py reports counter traffic only from a summary whose hash is this tree's."""
    import glob
    import hashlib

    h = hashlib.sha256()
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "csrc", "*"))) + [os.path.join(here, "Makefile")]:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
