"""raytracer_amd -- MI355X-native voxel-grid build + primary-ray grid traversal.

The product is ``lib/libhvrt.so`` (hand-written HIP for gfx950 behind the C ABI in ``include/hvrt.h``)
and the C++ host mirror of the reference's plug-in interface in ``host/``.  This Python package is a
thin ctypes binding used by the tests, ``bench.py`` and the multi-GPU driver; it contains no CPU
implementation of the path.
"""
from . import capi, meshgen  # noqa: F401
from .capi import Context, HvrtError, CameraFrame, make_camera_frame, default_camera, HVRT_MISS  # noqa: F401

__all__ = ["capi", "meshgen", "Context", "HvrtError", "CameraFrame", "make_camera_frame", "default_camera",
           "HVRT_MISS"]
