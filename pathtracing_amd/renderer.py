"""The drop-in boundary: ``trait Renderer`` (``worker/src/renderer/mod.rs:9-13``)
with the MI355X implementation behind it."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Protocol

import numpy as np

from . import _native as N
from .api import RenderOpts, RenderTask
from .scene import Scene


class Renderer(Protocol):
    """``fn init(scene) -> Self; async fn render(&mut self, task) -> Rgb32FImage``"""

    @classmethod
    def init(cls, scene: Scene) -> "Renderer": ...

    def render(self, render_task: RenderTask) -> np.ndarray: ...


class HipRenderer:
    """Counterpart of ``CPURenderer`` (``cpu_renderer/mod.rs:178-195``) running on one GPU."""

    def __init__(self, handle: C.c_void_p, device: int):
        self._handle = handle
        self.device = device
        self.last_stats: Optional[N.PtRenderStats] = None

    @classmethod
    def init(cls, scene: Scene, device: int = 0) -> "HipRenderer":
        return cls.from_desc(scene.desc_pointer, device)

    @classmethod
    def from_desc(cls, desc_pointer, device: int = 0) -> "HipRenderer":
        h = C.c_void_p()
        N.check(N.lib.pt_scene_create(desc_pointer, device, C.byref(h)))
        return cls(h, device)

    def render(self, render_task: RenderTask, opts: Optional[RenderOpts] = None) -> np.ndarray:
        """Returns the image as float32 ``[height, width, 3]``, top row first — the
        buffer ``get_image`` builds (``cpu_renderer/mod.rs:143-175``)."""
        t = render_task.to_native()
        w, h = int(t.camera.resolution_x), int(t.camera.resolution_y)
        out = np.empty((h, w, 3), dtype=np.float32)
        o = (opts or RenderOpts()).to_native()
        stats = N.PtRenderStats()
        N.check(N.lib.pt_render(self._handle, C.byref(t), C.byref(o), out.ctypes.data, C.byref(stats)))
        self.last_stats = stats
        return out

    def render_accumulate_device(self, render_task: RenderTask, opts: RenderOpts, accum_ptr: int, stream: int = 0,
                                 want_stats: bool = False):
        """Adds this call's samples into a device buffer (camera-space sums)."""
        t = render_task.to_native()
        o = opts.to_native()
        stats = N.PtRenderStats() if want_stats else None
        N.check(N.lib.pt_render_accumulate_device(self._handle, C.byref(t), C.byref(o), C.c_void_p(accum_ptr),
                                                  C.c_void_p(stream), C.byref(stats) if want_stats else None))
        return stats

    def trace_closest(self, rays: np.ndarray, brute_force: bool = False, route: str = "thread") -> np.ndarray:
        """``rays``: float32 ``[n, 8]`` (source, direction, min, max). Returns a
        structured array with the fields of ``PtHit``.  ``route``: ``"thread"`` (one
        thread per ray, every search in place), ``"wavefront"`` (the kernels a render
        uses: queue regions, listed searches, the search kernel, the walk) or
        ``"wavefront_split"`` (the same with the separate walk kernel)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        assert rays.ndim == 2 and rays.shape[1] == 8
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        flags = {"thread": 0, "wavefront": N.PT_FLAG_WAVEFRONT, "wavefront_split": N.PT_FLAG_WAVEFRONT_SPLIT}[route]
        if brute_force:
            flags |= N.PT_FLAG_BRUTE_FORCE
        N.check(N.lib.pt_trace_closest(self._handle, rays.ctypes.data, rays.shape[0], flags, hits.ctypes.data))
        return hits

    @property
    def info(self) -> N.PtSceneInfo:
        """What ``pt_scene_create`` did: .tree files adopted, structures built, times."""
        out = N.PtSceneInfo()
        N.check(N.lib.pt_scene_info(self._handle, C.byref(out)))
        return out

    def shade(self, directions: np.ndarray, hits: np.ndarray, stream_ids: np.ndarray, seed: int = 0):
        directions = np.ascontiguousarray(directions, dtype=np.float32)
        hits = np.ascontiguousarray(hits)
        stream_ids = np.ascontiguousarray(stream_ids, dtype=np.uint64)
        n = directions.shape[0]
        kind = np.zeros(n, dtype=np.uint32)
        a = np.zeros((n, 3), dtype=np.float32)
        d = np.zeros((n, 3), dtype=np.float32)
        N.check(N.lib.pt_shade(self._handle, directions.ctypes.data, hits.ctypes.data, stream_ids.ctypes.data, seed, n,
                               kind.ctypes.data, a.ctypes.data, d.ctypes.data))
        return kind, a, d

    def close(self) -> None:
        if self._handle:
            N.lib.pt_scene_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


HIT_DTYPE = np.dtype(
    [
        ("hit", np.uint32),
        ("hit_inside", np.uint32),
        ("node_index", np.uint32),
        ("tri_index", np.uint32),
        ("material_id", np.uint32),
        ("t", np.float32),
        ("point", np.float32, 3),
        ("normal", np.float32, 3),
        ("uv", np.float32, 2),
    ]
)
assert HIT_DTYPE.itemsize == C.sizeof(N.PtHit)


def generate_rays(camera, seed: int, pixel_xy: np.ndarray, sample_index: np.ndarray, device: int = 0) -> np.ndarray:
    """``Camera::get_ray`` (``camera.rs:51-83``) on the GPU for (pixel, sample) pairs."""
    cam = camera.to_native()
    pixel_xy = np.ascontiguousarray(pixel_xy, dtype=np.uint32)
    sample_index = np.ascontiguousarray(sample_index, dtype=np.uint64)
    n = sample_index.shape[0]
    rays = np.zeros((n, 8), dtype=np.float32)
    N.check(N.lib.pt_generate_rays(C.byref(cam), seed, pixel_xy.ctypes.data, sample_index.ctypes.data, n, device,
                                   rays.ctypes.data))
    return rays
