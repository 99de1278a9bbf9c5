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
        d = N.lib.pt_host_scene_desc(self._handle).contents
        d._owner = self  # the arrays it points into live as long as this Scene
        return d

    @property
    def desc_pointer(self):
        """Pointer to the PtSceneDesc; it keeps this Scene (which owns the arrays) alive."""
        p = N.lib.pt_host_scene_desc(self._handle)
        p._owner = self
        return p

    def cache_trees(self, root_dir: str) -> int:
        """Writes the ``.tree`` file of every ``kd_tree`` node whose file is missing,
        empty or stale (and takes it into this scene): the next ``Scene.load`` of the
        same files finds them and ``HipRenderer.init`` skips the build.  Returns the
        number of files written."""
        written = C.c_uint32()
        N.check(N.lib.pt_host_scene_cache_trees(self._handle, root_dir.encode(), C.byref(written)))
        return int(written.value)

    def close(self) -> None:
        if self._handle:
            N.lib.pt_host_scene_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_tree(mesh_desc: N.PtMeshDesc) -> bytes:
    """Serialised acceleration structure (".tree" contents) for one mesh of a scene."""
    data = C.POINTER(C.c_uint8)()
    size = C.c_uint64()
    N.check(N.lib.pt_tree_build(C.byref(mesh_desc), C.byref(data), C.byref(size)))
    try:
        return C.string_at(data, size.value)
    finally:
        N.lib.pt_buffer_free(data)


def validate_tree(mesh_desc: N.PtMeshDesc, blob: bytes) -> bool:
    buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob) if blob else (C.c_uint8 * 1)()
    return N.lib.pt_tree_validate(C.byref(mesh_desc), C.cast(buf, C.c_void_p), len(blob)) == N.PT_OK
