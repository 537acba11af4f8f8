"""raytracertest_amd -- MI355X-native drop-in for the RayTracer/ sub-project of
ipilter/RayTracerTest (per-pixel trace path only).

The compute lives in csrc/ (hand-written HIP for gfx950) behind the C ABI of
include/rt_mi355x.h; this package is the thin host mirror of rt::RayTracer
(RayTracer/RayTracer.h:14-41) over that ABI plus the row-band multi-GPU driver.
"""
from .api import RayTracer, RtError, load_library, library_path, device_count, MATH_FMA, MATH_STRICT  # noqa: F401
from . import scenes  # noqa: F401

__all__ = ["RayTracer", "RtError", "load_library", "library_path", "device_count", "scenes",
           "MATH_FMA", "MATH_STRICT"]
