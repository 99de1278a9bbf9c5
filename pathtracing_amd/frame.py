"""``Frame`` (``client/src/frame.rs:11-103``): what the reference client does with
the renders its workers send back — an element-wise running sum, the mean image
and its RGBA8 preview ``(c / (1 + c) * 255) as u8`` — kept on one GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N


class Frame:
    """``Frame::new(width, height)`` (``frame.rs:22-33``)."""

    def __init__(self, width: int, height: int, device: int = 0):
        self.width, self.height, self.device = int(width), int(height), device
        self._handle = C.c_void_p()
        N.check(N.lib.pt_frame_create(self.width, self.height, device, C.byref(self._handle)))

    @property
    def count(self) -> int:
        return int(N.lib.pt_frame_render_count(self._handle))

    def add_render(self, render: np.ndarray) -> None:
        """``RenderSum::add_render`` (``frame.rs:61-71``); ``render`` is float32
        ``[height, width, 3]`` as :meth:`HipRenderer.render` returns it.  The
        reference does not check the size (``frame.rs:36``); this does."""
        render = np.ascontiguousarray(render, dtype=np.float32)
        if render.shape != (self.height, self.width, 3):
            raise ValueError(f"render is {render.shape}, frame is {(self.height, self.width, 3)}")
        N.check(N.lib.pt_frame_add_render(self._handle, render.ctypes.data))

    def add_render_device(self, device_pointer: int, stream: int = 0) -> None:
        """The same for a render already in this GPU's memory (``width*height*3`` floats)."""
        N.check(N.lib.pt_frame_add_render_device(self._handle, C.c_void_p(device_pointer), C.c_void_p(stream)))

    def mean_image(self) -> np.ndarray:
        """``RenderSum::into_image`` (``frame.rs:73-83``): ``sum / max(count, 1)``."""
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        N.check(N.lib.pt_frame_image(self._handle, out.ctypes.data, None))
        return out

    def image(self) -> np.ndarray:
        """What ``Frame::add_render`` publishes (``frame.rs:44-46``): the
        tone-mapped mean as uint8 ``[height, width, 4]``."""
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        N.check(N.lib.pt_frame_image(self._handle, None, out.ctypes.data))
        return out

    def image_device(self, rgb_pointer: int = 0, rgba_pointer: int = 0, stream: int = 0) -> None:
        """Mean image (``width*height*3`` floats) and / or preview (``width*height*4``
        bytes) into this GPU's memory, enqueued on ``stream``; 0 skips an output."""
        N.check(N.lib.pt_frame_image_device(self._handle, C.c_void_p(rgb_pointer or None), C.c_void_p(rgba_pointer or None),
                                            C.c_void_p(stream)))

    def close(self) -> None:
        if self._handle:
            N.lib.pt_frame_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
