"""Scene loading: the host side of ``Scene::load`` (``worker/src/scene/mod.rs:58-112``),
done by the library's C++ loader and kept as an opaque handle."""
from __future__ import annotations

import ctypes as C

from . import _native as N


class Scene:
    """Host-resident scene = what the reference keeps in ``Arc<Scene>``."""

    def __init__(self, handle: int):
        self._handle = C.c_void_p(handle)

    @staticmethod
    def load(root_dir: str, scene_path: str) -> "Scene":
        """``root_dir`` plays the role of the GridFS bucket (``file_store.rs:12-35``):
        every path in the scene files is resolved against it."""
        h = C.c_void_p()
        N.check(N.lib.pt_host_scene_load(root_dir.encode(), scene_path.encode(), C.byref(h)))
        return Scene(h.value)

    @property
    def desc(self) -> N.PtSceneDesc:
        return N.lib.pt_host_scene_desc(self._handle).contents

    @property
    def desc_pointer(self):
        return N.lib.pt_host_scene_desc(self._handle)

    def close(self) -> None:
        if self._handle:
            N.lib.pt_host_scene_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
