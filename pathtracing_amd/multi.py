"""One process driving several GPUs of a node: ``pt_render_multi`` and the split it uses.

The reference combines workers on the host (the client sends the same task to every worker
and averages the frames, ``client/src/worker_pool.rs:221-236``, ``client/src/frame.rs:60-84``);
here the devices render their shares side by side and the sums are reduced onto the first
device (RCCL over xGMI).  ``pathtracing_amd.distributed`` is the one-process-per-GPU form of
the same split."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _native as N
from .api import RenderOpts, RenderTask
from .renderer import HipRenderer

_MODES = {"samples": N.SHARD_SAMPLES, "rows": N.SHARD_ROWS}


def shard(task: RenderTask, opts: Optional[RenderOpts], mode: str, world: int, rank: int) -> Tuple[int, int, Optional[tuple]]:
    """The library's own split (``pt_shard``): (iterations, sample_offset, tile or None) of
    ``rank``; iterations = 0 or a tile without rows means the rank has nothing to render."""
    t = task.to_native()
    o = (opts or RenderOpts()).to_native()
    ot, oo, active = N.PtRenderTask(), N.PtRenderOpts(), C.c_uint32()
    N.check(N.lib.pt_shard(C.byref(t), C.byref(o), _MODES[mode], world, rank, C.byref(ot), C.byref(oo), C.byref(active)))
    tile = (oo.tile_x0, oo.tile_y0, oo.tile_w, oo.tile_h) if mode == "rows" else None
    return (int(ot.iterations) if active.value or mode == "samples" else 0), int(oo.sample_offset), tile


def render_multi(renderers: Sequence[HipRenderer], task: RenderTask, opts: Optional[RenderOpts] = None, mode: str = "samples",
                 out_ptr: Optional[int] = None, width_height: Optional[Tuple[int, int]] = None):
    """``Renderer::render`` over the devices the ``renderers`` live on (one scene handle per
    device).  Returns (image float32 [H, W, 3] top row first, PtRenderStats).  With ``out_ptr``
    (W*H*3 floats in the memory of the first renderer's device) the image is resolved there,
    nothing is copied to the host and the first item returned is None."""
    t = task.to_native()
    o = (opts or RenderOpts()).to_native()
    w, h = int(t.camera.resolution_x), int(t.camera.resolution_y)
    handles = (C.c_void_p * len(renderers))(*[r._handle for r in renderers])
    stats = N.PtRenderStats()
    if out_ptr is not None:
        if width_height is not None and tuple(width_height) != (w, h):
            raise ValueError("the output buffer was sized for another resolution")
        N.check(N.lib.pt_render_multi(handles, len(renderers), C.byref(t), C.byref(o), _MODES[mode], C.c_void_p(out_ptr),
                                      C.byref(stats)))
        return None, stats
    out = np.empty((h, w, 3), dtype=np.float32)
    N.check(N.lib.pt_render_multi(handles, len(renderers), C.byref(t), C.byref(o), _MODES[mode], out.ctypes.data, C.byref(stats)))
    return out, stats
