"""MI355X-native path-tracing core behind the reference's ``Renderer`` boundary.

Python here is a thin host layer (ctypes) over ``libpathtracing_amd.so``; the
compute path is the HIP library and nothing else.
"""
from .api import Camera, Config, RenderOpts, RenderTask
from .frame import Frame
from .multi import render_multi
from .renderer import HIT_DTYPE, HipRenderer, Renderer, generate_rays
from .scene import Scene
from .worker import RenderedImage, Worker

__all__ = [
    "Camera",
    "Config",
    "RenderOpts",
    "RenderTask",
    "Frame",
    "HipRenderer",
    "Renderer",
    "Scene",
    "Worker",
    "RenderedImage",
    "generate_rays",
    "render_multi",
    "HIT_DTYPE",
]
