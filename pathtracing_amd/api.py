"""Render-task types, mirroring ``worker/src/api/render_task.rs:5-49`` and
``worker/src/camera.rs:8-48`` of the reference (same JSON schema)."""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass, field
from typing import List, Optional

from . import _native as N

_BOKEH = {"point": N.BOKEH_POINT, "circle": N.BOKEH_CIRCLE, "square": N.BOKEH_SQUARE}
_BOKEH_NAME = {v: k for k, v in _BOKEH.items()}


@dataclass
class Config:
    """api/render_task.rs:45-49"""

    trace_depth: int
    iterations: int


@dataclass
class Camera:
    """camera.rs:35-48 (``fov`` is the JSON field ``field_of_view``)."""

    resolution: List[int]
    rotation: List[List[float]]
    position: List[float]
    fov: float
    near_plane: float
    focal_length: float
    bokeh_shape: str
    bokeh_size: float

    def to_native(self) -> N.PtCamera:
        c = N.PtCamera()
        c.resolution_x, c.resolution_y = int(self.resolution[0]), int(self.resolution[1])
        for r, name in enumerate(("row0", "row1", "row2")):
            row = getattr(c.rotation, name)
            row.x, row.y, row.z = (float(v) for v in self.rotation[r])
        c.position.x, c.position.y, c.position.z = (float(v) for v in self.position)
        c.fov = float(self.fov)
        c.near_plane = float(self.near_plane)
        c.focal_length = float(self.focal_length)
        if self.bokeh_shape not in _BOKEH:
            raise ValueError(f"unknown bokeh_shape {self.bokeh_shape!r}")
        c.bokeh_shape = _BOKEH[self.bokeh_shape]
        c.bokeh_size = float(self.bokeh_size)
        return c


@dataclass
class RenderTask:
    """api/render_task.rs:23-29.  ``scene_md5`` is empty for the file form
    (``RenderTaskUninit``, :5-10) that ``scene_data/render_task.json`` holds."""

    scene: str
    config: Config
    camera: Camera
    scene_md5: str = ""

    @staticmethod
    def from_json(text: str) -> "RenderTask":
        """Parsed by the library's own reader so Python and C++ callers agree."""
        task = N.PtRenderTask()
        buf = C.create_string_buffer(4096)
        N.check(N.lib.pt_render_task_parse(text.encode("utf-8"), C.byref(task), buf, len(buf)))
        cam = task.camera
        rot = [[getattr(cam.rotation, r).x, getattr(cam.rotation, r).y, getattr(cam.rotation, r).z] for r in ("row0", "row1", "row2")]
        md5 = json.loads(text).get("scene_md5", "")
        return RenderTask(
            scene=buf.value.decode("utf-8"),
            scene_md5=md5,
            config=Config(trace_depth=int(task.trace_depth), iterations=int(task.iterations)),
            camera=Camera(
                resolution=[int(cam.resolution_x), int(cam.resolution_y)],
                rotation=rot,
                position=[cam.position.x, cam.position.y, cam.position.z],
                fov=cam.fov,
                near_plane=cam.near_plane,
                focal_length=cam.focal_length,
                bokeh_shape=_BOKEH_NAME[int(cam.bokeh_shape)],
                bokeh_size=cam.bokeh_size,
            ),
        )

    def to_native(self) -> N.PtRenderTask:
        t = N.PtRenderTask()
        t.camera = self.camera.to_native()
        t.trace_depth = int(self.config.trace_depth)
        t.iterations = int(self.config.iterations)
        return t


@dataclass
class RenderOpts:
    """Knobs the reference does not have (all default to "like the reference")."""

    seed: int = 0
    sample_offset: int = 0
    brute_force: bool = False
    tile: Optional[tuple] = None  # (x0, y0, w, h) in camera space
    paths_per_batch: int = 0
    profile: bool = False  # time each BVH search kernel launch (RenderStats.search_kernel_ms)
    single_stream: bool = False  # never keep two batches in flight
    count_visits: bool = False  # count the node / triangle records the search kernel fetches (RenderStats)

    def to_native(self) -> N.PtRenderOpts:
        o = N.PtRenderOpts()
        o.seed = self.seed
        o.sample_offset = self.sample_offset
        o.flags = ((N.PT_FLAG_BRUTE_FORCE if self.brute_force else 0) | (N.PT_FLAG_PROFILE if self.profile else 0)
                   | (N.PT_FLAG_SINGLE_STREAM if self.single_stream else 0)
                   | (N.PT_FLAG_COUNT_VISITS if self.count_visits else 0))
        if self.tile is not None:
            o.tile_x0, o.tile_y0, o.tile_w, o.tile_h = (int(v) for v in self.tile)
        o.paths_per_batch = self.paths_per_batch
        return o
