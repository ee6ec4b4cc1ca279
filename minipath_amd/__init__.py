"""minipath_amd -- MI355X (gfx950) implementation of bluecube/minipath's per-pixel sampling hot path.

Host-side mirror of the reference's public API (src/lib.rs:8-10): ``render``, ``RenderProgress``,
``RenderSettings``, ``Camera``, ``Scene`` and ``TriangleBvh``, over the C ABI of ``include/minipath_hip.h``.
All compute happens in ``csrc/libminipath_hip.so`` (hand-written HIP kernels); there is no CPU path here.
"""
from ._lib import MinipathError, MP_NO_PRIM, MP_LINK_NULL, MAX_MATERIALS, SO_PATH  # noqa: F401
from .camera import Camera, CameraSampler  # noqa: F401
from .screen_block import ScreenBlock, tile_ordering  # noqa: F401
from .scene import Context, Instances, ObjectGroup, Scene, Sphere, TriangleBvh  # noqa: F401
from .renderer import (RenderProgress, RenderProgressSnapshot, RenderSettings, render, render_multi, render_tile,  # noqa: F401
                       FrameRenderer, MultiDeviceFrame)

__all__ = [
    "Camera", "CameraSampler", "Context", "Instances", "FrameRenderer", "MinipathError", "MultiDeviceFrame", "ObjectGroup", "RenderProgress", "render_multi",
    "RenderProgressSnapshot", "RenderSettings", "Scene", "ScreenBlock", "Sphere", "TriangleBvh", "render", "render_tile",
    "tile_ordering",
]
