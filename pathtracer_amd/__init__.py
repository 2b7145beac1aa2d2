"""pathtracer_amd - MI355X-native per-ray hot path of felpzOliveira/PathTracer.

csrc/   HIP kernels (gfx950) + the C-ABI of include/ptr_api.h  -> lib/libptr_hip.so
host/   C++ mirror of the reference's scene-description API       -> lib/libpt_host.so
capi    ctypes view of both; scene / render: Python drivers over the C-ABI
"""
from . import assets, capi  # noqa: F401
from .capi import (LI_DIRECT, LI_PATH, LI_PATHSAMPLED, LI_VOLPATH, TONEMAP_EXPONENTIAL, TONEMAP_NAUGHTYDOG,  # noqa: F401
                   TONEMAP_REINHARD, PtrError)  # noqa: F401
from .render import Renderer, set_bvh_build_mode  # noqa: F401
from .scene import BuiltScene, FlatSceneBuilder, ImageTexture, look_at_camera, material, medium  # noqa: F401
