"""Host-side driver of the render path, mirroring the reference's render() sequence
(src/main.cpp:1092-1157: PrepareSceneForRendering -> SetupPixels -> Render per sample batch ->
read the Pixel array), every step going through the C-ABI of include/ptr_api.h.
"""
import ctypes as C

import numpy as np

from . import capi


def set_bvh_build_mode(mode):
    """'host' (binned SAH, default) or 'device' (LBVH built on the current GPU): who builds the geometry BVHs of the
    scenes uploaded from now on (ptr_scene_build_mode, process-wide)."""
    capi.check(capi.hip().ptr_scene_build_mode({"host": 0, "device": 1}[mode]))


class Renderer:
    def __init__(self, device=-1):
        self._ctx = capi.CTX()
        capi.check(capi.hip().ptr_create(int(device), C.byref(self._ctx)))
        self.width = self.height = 0

    # -- lifetime ---------------------------------------------------------------------
    def close(self):
        if self._ctx:
            capi.hip().ptr_destroy(self._ctx)
            self._ctx = capi.CTX()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def ctx(self):
        return self._ctx

    def set_stream(self, hip_stream):
        capi.check(capi.hip().ptr_set_stream(self._ctx, C.c_void_p(int(hip_stream))))

    # -- scene ------------------------------------------------------------------------
    def upload(self, scene, srand=1):
        """scene: BuiltScene / PyScene (anything with a .flat POINTER(PtrFlatScene)).

        Like the reference's Aggregator::Wrap, ptr_scene_upload draws from the process's libc
        rand() (primitive.cpp:541) - it fixes the order in which primitives are visited.  srand
        (default 1 = a fresh process) reseeds libc first so that an upload is reproducible within
        a long-lived Python process; pass None to leave the host application's rand() state alone."""
        if srand is not None:
            libc = C.CDLL(None)
            libc.srand(C.c_uint(int(srand)))
            for _ in range(int(getattr(scene, "rand_calls", 0))):
                libc.rand()
        capi.check(capi.hip().ptr_scene_upload(self._ctx, scene.flat))

    def set_camera(self, cam):
        capi.check(capi.hip().ptr_camera_set(self._ctx, C.byref(cam)))

    def film_init(self, width, height, seed=1):
        capi.check(capi.hip().ptr_film_init(self._ctx, int(width), int(height), C.c_uint64(int(seed))))
        self.width, self.height = int(width), int(height)

    def shard(self, offset, stride):
        capi.check(capi.hip().ptr_film_shard(self._ctx, int(offset), int(stride)))

    # -- render -----------------------------------------------------------------------
    def tiles(self, offset, stride):
        """ptr_film_tiles: this context renders the 8x8-pixel tiles t with t % stride == offset (call after film_init)."""
        capi.check(capi.hip().ptr_film_tiles(self._ctx, int(offset), int(stride)))

    def render(self, ns, integrator=capi.LI_VOLPATH):
        capi.check(capi.hip().ptr_render(self._ctx, int(ns), int(integrator)))

    def sync(self):
        capi.check(capi.hip().ptr_sync(self._ctx))

    def film(self):
        """(we[h,w,3] float32, acc_weight[h,w] float64, samples[h,w] int64)"""
        n = self.width * self.height
        we = np.empty((n, 3), dtype=np.float32)
        acc = np.empty(n, dtype=np.float64)
        smp = np.empty(n, dtype=np.int64)
        capi.check(capi.hip().ptr_film_read(self._ctx, we.ctypes.data, acc.ctypes.data, smp.ctypes.data))
        sh = (self.height, self.width)
        return we.reshape(sh + (3,)), acc.reshape(sh), smp.reshape(sh)

    def film_write(self, we, acc, samples, next_sample=-1):
        """Restore saved accumulators (ptr_film_write); rendering continues at next_sample."""
        n = self.width * self.height
        we = np.ascontiguousarray(we, dtype=np.float32).reshape(n, 3)
        acc = np.ascontiguousarray(acc, dtype=np.float64).reshape(n)
        samples = np.ascontiguousarray(samples, dtype=np.int64).reshape(n)
        capi.check(capi.hip().ptr_film_write(self._ctx, we.ctypes.data, acc.ctypes.data, samples.ctypes.data, int(next_sample)))

    def rgb8(self, exposure=2.0, algorithm=capi.TONEMAP_EXPONENTIAL):
        """GetPixelRGB over the film on the device (camera.cpp:36-51): uint8 [h, w, 3], row 0 = bottom
        of the frame like the Pixel array.  Defaults are render()'s (main.cpp:1185)."""
        out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        capi.check(capi.hip().ptr_film_rgb8(self._ctx, float(exposure), int(algorithm), out.ctypes.data))
        return out

    def save(self, path, exposure=2.0, algorithm=capi.TONEMAP_EXPONENTIAL):
        """ImageWrite (camera.cpp:53-85): PNG, or binary PPM when the path ends in .ppm (ppm.cpp:4-32)."""
        rc = capi.host().pth_image_write(self._ctx, self.width, self.height, str(path).encode(), float(exposure), int(algorithm))
        if rc == 1:
            raise capi.PtrError(rc, capi.hip().ptr_last_error().decode("utf-8", "replace"))
        if rc:
            raise OSError("could not write %s (code %d)" % (path, rc))

    def checkpoint(self, path):
        """Float accumulators to one .npz-like file (np.savez): resume with Renderer.resume."""
        we, acc, smp = self.film()
        with open(path, "wb") as f:
            np.savez(f, we=we, acc=acc, samples=smp)

    def resume(self, path, next_sample=-1):
        with np.load(path, allow_pickle=False) as z:
            if z["acc"].shape != (self.height, self.width):
                raise ValueError("checkpoint is %s, film is %s" % (z["acc"].shape, (self.height, self.width)))
            self.film_write(z["we"], z["acc"], z["samples"], next_sample)

    def film_device_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        capi.check(capi.hip().ptr_film_device_ptrs(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def tiles_packed_pixels(self, stride):
        """entries in the packed film of each of `stride` tile shards (ptr_film_tiles_packed_pixels)"""
        n = C.c_int64(0)
        capi.check(capi.hip().ptr_film_tiles_packed_pixels(self._ctx, int(stride), C.byref(n)))
        return int(n.value)

    def tiles_pack(self, d_we, d_acc, d_samples):
        """this context's own tiles into packed device buffers (ptr_film_tiles_pack); asynchronous on the context's stream"""
        capi.check(capi.hip().ptr_film_tiles_pack(self._ctx, d_we, d_acc, d_samples))

    def tiles_unpack(self, offset, stride, d_we, d_acc, d_samples, d_out_we, d_out_acc, d_out_samples):
        """the packed film of shard (offset, stride) into full-frame device accumulators (ptr_film_tiles_unpack)"""
        capi.check(capi.hip().ptr_film_tiles_unpack(self._ctx, int(offset), int(stride), d_we, d_acc, d_samples, d_out_we, d_out_acc, d_out_samples))

    def fill_samples(self, d_out_samples, value):
        capi.check(capi.hip().ptr_film_fill_samples(self._ctx, d_out_samples, int(value)))

    def image(self):
        """GetPixelRGB (camera.cpp:36-51): we / accWeight, NaN -> 0."""
        we, acc, _ = self.film()
        with np.errstate(divide="ignore", invalid="ignore"):
            img = we / acc[..., None].astype(np.float32)
        img[~np.isfinite(img)] = 0
        return img

    # -- stages -----------------------------------------------------------------------
    def generate_rays(self, sample, d_o, d_d, d_weight=None):
        capi.check(capi.hip().ptr_generate_rays(self._ctx, int(sample), d_o, d_d, d_weight))

    def trace_closest(self, n, d_o, d_d, d_tmax, d_prim, d_face, d_t, d_b0=None, d_b1=None):
        capi.check(capi.hip().ptr_trace_closest(self._ctx, int(n), d_o, d_d, d_tmax, d_prim, d_face, d_t, d_b0, d_b1))

    def trace_closest_noalloc(self, n, d_o, d_d, d_tmax, d_prim, d_face, d_t, d_b0=None, d_b1=None):
        """ptr_trace_closest_noalloc: same results, no device allocation, cursor in the library's static scratch."""
        capi.check(capi.hip().ptr_trace_closest_noalloc(self._ctx, int(n), d_o, d_d, d_tmax, d_prim, d_face, d_t, d_b0, d_b1))

    def trace_occluded(self, n, d_o, d_d, d_tmax, d_occ):
        capi.check(capi.hip().ptr_trace_occluded(self._ctx, int(n), d_o, d_d, d_tmax, d_occ))

    # -- unit hooks (known-answer tests) ------------------------------------------------
    UNIT_KINDS = {"fr_dielectric": (0, 3, 1), "fr_conductor": (1, 10, 3), "ggx_d": (2, 5, 1), "ggx_g1": (3, 5, 1), "ggx_pdf": (4, 8, 1),
                  "ggx_sample_wh": (5, 7, 3), "hg_p": (6, 2, 1), "hg_sample": (7, 6, 4), "rough_to_alpha": (8, 1, 1), "env_sample": (9, 2, 3),
                  "env_pdf": (10, 2, 1), "sss_sr": (11, 1, 3), "sss_pdf_sr": (12, 2, 1), "sss_sample_sr": (13, 2, 1)}

    def unit_eval(self, kind, rows, prim=-1):
        """ptr_unit_eval over host rows [n, in width] -> numpy [n, out width] (copies through torch tensors on this device)"""
        import numpy as np
        import torch

        kid, iw, ow = self.UNIT_KINDS[kind]
        a = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, iw)).cuda()
        out = torch.zeros((a.shape[0], ow), dtype=torch.float32, device=a.device)
        torch.cuda.synchronize()
        capi.check(capi.hip().ptr_unit_eval(self._ctx, int(prim), kid, a.shape[0], a.data_ptr(), iw, out.data_ptr(), ow))
        self.sync()
        return out.cpu().numpy()

    def unit_bsdf(self, prim, wo, wi=None, u2=None, uv=None):
        """ptr_unit_bsdf: (f, pdf) for directions wi, or (wi, f, pdf, flags) for sample pairs u2"""
        import numpy as np
        import torch

        two = torch.from_numpy(np.ascontiguousarray(wo, dtype=np.float32).reshape(-1, 3)).cuda()
        n = two.shape[0]
        tuv = torch.from_numpy(np.full((n, 2), 0.5, np.float32) if uv is None else np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)).cuda()
        f = torch.zeros((n, 3), dtype=torch.float32, device=two.device)
        pdf = torch.zeros(n, dtype=torch.float32, device=two.device)
        wout = torch.zeros((n, 3), dtype=torch.float32, device=two.device)
        fl = torch.zeros(n, dtype=torch.int32, device=two.device)
        sample = wi is None
        x = torch.from_numpy(np.ascontiguousarray(u2 if sample else wi, dtype=np.float32).reshape(n, -1)).cuda()
        torch.cuda.synchronize()
        capi.check(capi.hip().ptr_unit_bsdf(self._ctx, int(prim), 1 if sample else 0, n, two.data_ptr(), x.data_ptr(), tuv.data_ptr(), wout.data_ptr(),
                                            f.data_ptr(), pdf.data_ptr(), fl.data_ptr()))
        self.sync()
        if sample:
            return wout.cpu().numpy(), f.cpu().numpy(), pdf.cpu().numpy(), fl.cpu().numpy()
        return f.cpu().numpy(), pdf.cpu().numpy()

    # -- introspection ----------------------------------------------------------------
    def counters(self):
        c = capi.PtrCounters()
        capi.check(capi.hip().ptr_counters_read(self._ctx, C.byref(c)))
        return c.as_dict()

    def pipeline_stats(self):
        st = capi.PtrPipelineStats()
        capi.check(capi.hip().ptr_pipeline_stats_read(self._ctx, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in st._fields_}

    def counters_reset(self):
        capi.check(capi.hip().ptr_counters_reset(self._ctx))

    def profile_enable(self, timing=True, count_traversal=False):
        capi.check(capi.hip().ptr_profile_enable(self._ctx, (1 if timing else 0) | (2 if count_traversal else 0)))

    def profile_reset(self):
        capi.check(capi.hip().ptr_profile_reset(self._ctx))

    def profile(self):
        arr = (capi.PtrKernelProfile * capi.PROF_KERNELS)()
        n = C.c_int(0)
        capi.check(capi.hip().ptr_profile_read(self._ctx, arr, capi.PROF_KERNELS, C.byref(n)))
        return {arr[i].name.decode(): {"launches": int(arr[i].launches), "total_ms": float(arr[i].total_ms), "items": int(arr[i].items)}
                for i in range(n.value)}

    def bvh_build_info(self):
        o = capi.PtrBvhBuildInfo()
        capi.check(capi.hip().ptr_bvh_build_info(self._ctx, C.byref(o)))
        return capi.bvh_build_info_dict(o)

    def bvh_info(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        capi.check(capi.hip().ptr_bvh_info(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
        return {"nodes": a.value, "items": b.value, "bytes": c.value}
