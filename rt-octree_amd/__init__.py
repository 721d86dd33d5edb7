"""rt-octree_amd -- MI355X-native RT-Octree render path (host-side mirror of the reference's
operator interface over the C ABI in include/rto.h).

The directory name carries a hyphen (it is the name the build contract asks for); import it as
`rt_octree_amd` through the alias module at the repository root.
"""
from ._lib import LIB_PATH, RtoError, build_library, lib  # noqa: F401
from .volrend import (  # noqa: F401
    Camera, N3Tree, RenderContext, RenderOptions, Timer, launch_renderer, launch_renderer_batch, filtering,
    SUPPORTED_SPP, KERNEL_AUTO, KERNEL_GENERIC, KERNEL_FAST, FILTER_EXACT, FILTER_FAST,
)

__all__ = [
    "LIB_PATH", "RtoError", "build_library", "lib", "Camera", "N3Tree", "RenderContext",
    "RenderOptions", "Timer", "launch_renderer", "launch_renderer_batch", "filtering", "SUPPORTED_SPP",
    "KERNEL_AUTO", "KERNEL_GENERIC", "KERNEL_FAST", "FILTER_EXACT", "FILTER_FAST",
]
