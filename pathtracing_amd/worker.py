"""``Worker`` (``worker/src/lib.rs:18-49``): scene cache keyed by ``scene_md5`` and a
fresh renderer per task, plus the ``RenderedImage`` wire codec (``lib.rs:51-79``)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import _native as N
from .api import RenderOpts, RenderTask
from .renderer import HipRenderer
from .scene import Scene


class Worker:
    """``scene_root`` replaces the MongoDB URL: scene files are read from that directory."""

    def __init__(self, scene_root: str, device: int = 0, cache_trees: bool = True):
        self.scene_root = scene_root
        self.device = device
        self.cache_trees = cache_trees  # write missing / stale .tree files back into scene_root
        self.cached_scenes: Dict[str, Scene] = {}
        self._renderers: Dict[str, HipRenderer] = {}

    def render(self, render_task: RenderTask, opts: Optional[RenderOpts] = None) -> np.ndarray:
        key = render_task.scene_md5 or render_task.scene
        if key not in self.cached_scenes:
            self.cached_scenes[key] = Scene.load(self.scene_root, render_task.scene)
            if self.cache_trees:
                try:
                    self.cached_scenes[key].cache_trees(self.scene_root)
                except N.PathTracingError as e:  # a read-only scene directory is no reason to refuse the task
                    if e.status != 3:  # PT_ERR_IO
                        raise
        # The reference constructs a renderer per task (lib.rs:44); the device copy of the
        # scene is kept beside the cached host scene so the upload happens once.
        if key not in self._renderers:
            self._renderers[key] = HipRenderer.init(self.cached_scenes[key], self.device)
        return self._renderers[key].render(render_task, opts)


class RenderedImage:
    """``u32 LE width, u32 LE height, width*height*3 f32 LE`` (``lib.rs:56-78``)."""

    def __init__(self, image: np.ndarray):
        assert image.ndim == 3 and image.shape[2] == 3
        self.image = np.ascontiguousarray(image, dtype=np.float32)

    def to_bytes(self) -> bytes:
        h, w, _ = self.image.shape
        out = np.empty(8 + w * h * 12, dtype=np.uint8)
        N.check(N.lib.pt_rendered_image_to_bytes(self.image.ctypes.data, w, h, out.ctypes.data, out.size))
        return out.tobytes()

    @staticmethod
    def from_bytes(data: bytes) -> "RenderedImage":
        raw = np.frombuffer(data, dtype=np.uint8)
        w, h = C.c_uint32(), C.c_uint32()
        N.check(N.lib.pt_rendered_image_from_bytes(raw.ctypes.data, raw.size, C.byref(w), C.byref(h), None, 0))
        img = np.empty((h.value, w.value, 3), dtype=np.float32)
        N.check(N.lib.pt_rendered_image_from_bytes(raw.ctypes.data, raw.size, C.byref(w), C.byref(h), img.ctypes.data, img.size))
        return RenderedImage(img)
