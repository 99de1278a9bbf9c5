"""ctypes binding of ``libpathtracing_amd.so`` (declared in ``include/pathtracing_amd.h``).

The library is the product: there is no Python or CPU fallback.  If the shared
object has not been built, importing this module raises ``ImportError`` with the
build command, and every compute entry point needs a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PATHTRACING_AMD_LIB selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("PATHTRACING_AMD_LIB") or os.path.join(_HERE, "libpathtracing_amd.so")

PT_OK = 0
PT_NO_INDEX = 0xFFFFFFFF
PT_FLAG_BRUTE_FORCE = 1
PT_FLAG_PROFILE = 2
PT_FLAG_WAVEFRONT = 4
PT_FLAG_WAVEFRONT_SPLIT = 8
PT_FLAG_SINGLE_STREAM = 16
PT_FLAG_COUNT_VISITS = 32
PT_STATIC_TRACE_MAX_RAYS = 16777216

NODE_COLLECTION, NODE_TRANSFORM, NODE_MESH, NODE_SPHERE, NODE_PLANE, NODE_KD_TREE = range(6)
MATERIAL_BASE, MATERIAL_PBR = 0, 1
UV_CLAMP, UV_REPEAT = 0, 1
BOKEH_POINT, BOKEH_CIRCLE, BOKEH_SQUARE = 0, 1, 2
SHARD_SAMPLES, SHARD_ROWS = 0, 1

STATUS_NAMES = {
    0: "PT_OK",
    1: "PT_ERR_INVALID_ARGUMENT",
    2: "PT_ERR_SCENE",
    3: "PT_ERR_IO",
    4: "PT_ERR_PARSE",
    5: "PT_ERR_DEVICE",
    6: "PT_ERR_UNSUPPORTED",
}


class PtVec2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class PtVec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class PtMat3(C.Structure):
    _fields_ = [("row0", PtVec3), ("row1", PtVec3), ("row2", PtVec3)]


class PtMat4(C.Structure):
    _fields_ = [("m", (C.c_float * 4) * 4)]


class PtRay(C.Structure):
    _fields_ = [("source", PtVec3), ("direction", PtVec3), ("min", C.c_float), ("max", C.c_float)]


class PtNodeDesc(C.Structure):
    _fields_ = [
        ("type", C.c_uint32),
        ("child_count", C.c_uint32),
        ("matrix", PtMat4),
        ("mesh_id", C.c_uint32),
        ("material_id", C.c_uint32),
        ("center", PtVec3),
        ("radius", C.c_float),
        ("point", PtVec3),
        ("normal", PtVec3),
        ("tangent", PtVec3),
        ("bitangent", PtVec3),
        ("tree_id", C.c_uint32),
    ]


class PtMeshDesc(C.Structure):
    _fields_ = [
        ("triangle_count", C.c_uint64),
        ("positions", C.POINTER(C.c_float)),
        ("normals", C.POINTER(C.c_float)),
        ("normal_valid", C.POINTER(C.c_uint8)),
        ("uvs", C.POINTER(C.c_float)),
    ]


class PtMaterialDesc(C.Structure):
    _fields_ = [
        ("type", C.c_uint32),
        ("color_is_texture", C.c_uint32),
        ("color", PtVec3),
        ("image_id", C.c_uint32),
        ("uv_mode", C.c_uint32),
        ("emission", PtVec3),
        ("refraction", C.c_float),
        ("reflective", C.c_float),
        ("emissive", C.c_float),
        ("refractive", C.c_float),
        ("albedo", PtVec3),
        ("roughness", C.c_float),
        ("metallic", C.c_float),
    ]


class PtImageDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba8", C.POINTER(C.c_uint8))]


class PtTreeBlob(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("size", C.c_uint64)]


class PtSceneDesc(C.Structure):
    _fields_ = [
        ("nodes", C.POINTER(PtNodeDesc)),
        ("node_count", C.c_uint32),
        ("meshes", C.POINTER(PtMeshDesc)),
        ("mesh_count", C.c_uint32),
        ("materials", C.POINTER(PtMaterialDesc)),
        ("material_count", C.c_uint32),
        ("images", C.POINTER(PtImageDesc)),
        ("image_count", C.c_uint32),
        ("trees", C.POINTER(PtTreeBlob)),
        ("tree_count", C.c_uint32),
    ]


class PtSceneInfo(C.Structure):
    _fields_ = [
        ("mesh_count", C.c_uint32),
        ("trees_offered", C.c_uint32),
        ("trees_adopted", C.c_uint32),
        ("trees_built", C.c_uint32),
        ("instance_run_children", C.c_uint32),
        ("triangle_count", C.c_uint64),
        ("bvh_bytes", C.c_uint64),
        ("build_ms", C.c_double),
        ("setup_ms", C.c_double),
    ]


class PtCamera(C.Structure):
    _fields_ = [
        ("resolution_x", C.c_uint64),
        ("resolution_y", C.c_uint64),
        ("rotation", PtMat3),
        ("position", PtVec3),
        ("fov", C.c_float),
        ("near_plane", C.c_float),
        ("focal_length", C.c_float),
        ("bokeh_shape", C.c_uint32),
        ("bokeh_size", C.c_float),
    ]


class PtRenderTask(C.Structure):
    _fields_ = [("camera", PtCamera), ("trace_depth", C.c_uint64), ("iterations", C.c_uint64)]


class PtRenderOpts(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("sample_offset", C.c_uint64),
        ("flags", C.c_uint32),
        ("tile_x0", C.c_uint32),
        ("tile_y0", C.c_uint32),
        ("tile_w", C.c_uint32),
        ("tile_h", C.c_uint32),
        ("paths_per_batch", C.c_uint32),
    ]


class PtRenderStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("paths", C.c_uint64),
        ("render_ms", C.c_double),
        ("bvh_bytes", C.c_uint64),
        ("searches", C.c_uint64),
        ("search_kernel_ms", C.c_double),
        ("search_kernel_launches", C.c_uint64),
        ("node_visits", C.c_uint64),
        ("triangle_tests", C.c_uint64),
        ("wave_node_steps", C.c_uint64),
        ("wave_leaf_steps", C.c_uint64),
        ("wave_leaf_rounds", C.c_uint64),
        ("reduce_ms", C.c_double),
    ]


class PtHit(C.Structure):
    _fields_ = [
        ("hit", C.c_uint32),
        ("hit_inside", C.c_uint32),
        ("node_index", C.c_uint32),
        ("tri_index", C.c_uint32),
        ("material_id", C.c_uint32),
        ("t", C.c_float),
        ("point", PtVec3),
        ("normal", PtVec3),
        ("uv", PtVec2),
    ]


# Every symbol include/pathtracing_amd.h declares: name -> (restype, argtypes)
_VOIDP = C.c_void_p
class PtServerStats(C.Structure):
    _fields_ = [
        ("connections", C.c_uint64),
        ("handshakes", C.c_uint64),
        ("messages_answered", C.c_uint64),
        ("messages_failed", C.c_uint64),
        ("datagrams_answered", C.c_uint64),
    ]


SYMBOLS = {
    "pt_last_error": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "pt_abi_version": (C.c_uint32, []),
    "pt_scene_create": (C.c_int, [C.POINTER(PtSceneDesc), C.c_int, C.POINTER(_VOIDP)]),
    "pt_scene_destroy": (None, [_VOIDP]),
    "pt_scene_info": (C.c_int, [_VOIDP, C.POINTER(PtSceneInfo)]),
    "pt_device_release_workspace": (C.c_int, [C.c_int]),
    "pt_render": (C.c_int, [_VOIDP, C.POINTER(PtRenderTask), C.POINTER(PtRenderOpts), _VOIDP, C.POINTER(PtRenderStats)]),
    "pt_render_accumulate_device": (
        C.c_int,
        [_VOIDP, C.POINTER(PtRenderTask), C.POINTER(PtRenderOpts), _VOIDP, _VOIDP, C.POINTER(PtRenderStats)],
    ),
    "pt_resolve_device": (C.c_int, [_VOIDP, C.c_uint64, C.c_uint64, C.c_uint64, _VOIDP, _VOIDP]),
    "pt_shard": (C.c_int, [C.POINTER(PtRenderTask), C.POINTER(PtRenderOpts), C.c_uint32, C.c_uint32, C.c_uint32,
                          C.POINTER(PtRenderTask), C.POINTER(PtRenderOpts), C.POINTER(C.c_uint32)]),
    "pt_render_multi": (C.c_int, [C.POINTER(_VOIDP), C.c_int, C.POINTER(PtRenderTask), C.POINTER(PtRenderOpts), C.c_uint32,
                                 _VOIDP, C.POINTER(PtRenderStats)]),
    "pt_trace_closest": (C.c_int, [_VOIDP, _VOIDP, C.c_size_t, C.c_uint32, _VOIDP]),
    "pt_trace_closest_device_static": (C.c_int, [_VOIDP, _VOIDP, C.c_size_t, C.c_uint32, _VOIDP, _VOIDP]),
    "pt_generate_rays": (C.c_int, [C.POINTER(PtCamera), C.c_uint64, _VOIDP, _VOIDP, C.c_size_t, C.c_int, _VOIDP]),
    "pt_shade": (C.c_int, [_VOIDP, _VOIDP, _VOIDP, _VOIDP, C.c_uint64, C.c_size_t, _VOIDP, _VOIDP, _VOIDP]),
    "pt_host_scene_load": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_VOIDP)]),
    "pt_host_scene_desc": (C.POINTER(PtSceneDesc), [_VOIDP]),
    "pt_host_scene_destroy": (None, [_VOIDP]),
    "pt_host_scene_cache_trees": (C.c_int, [_VOIDP, C.c_char_p, C.POINTER(C.c_uint32)]),
    "pt_render_task_parse": (C.c_int, [C.c_char_p, C.POINTER(PtRenderTask), C.c_char_p, C.c_size_t]),
    "pt_tree_build": (C.c_int, [C.POINTER(PtMeshDesc), C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)]),
    "pt_buffer_free": (None, [C.POINTER(C.c_uint8)]),
    "pt_tree_validate": (C.c_int, [C.POINTER(PtMeshDesc), _VOIDP, C.c_uint64]),
    "pt_rendered_image_to_bytes": (C.c_int, [_VOIDP, C.c_uint32, C.c_uint32, _VOIDP, C.c_size_t]),
    "pt_rendered_image_from_bytes": (
        C.c_int,
        [_VOIDP, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _VOIDP, C.c_size_t],
    ),
    "pt_discovery_request_encode": (C.c_int, [_VOIDP, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pt_discovery_request_decode": (C.c_int, [_VOIDP, C.c_size_t]),
    "pt_discovery_response_encode": (C.c_int, [C.c_uint16, _VOIDP, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pt_discovery_response_decode": (C.c_int, [_VOIDP, C.c_size_t, C.POINTER(C.c_uint16)]),
    "pt_scene_md5": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p]),
    "pt_worker_create": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_VOIDP)]),
    "pt_worker_destroy": (None, [_VOIDP]),
    "pt_worker_cached_scene_count": (C.c_uint32, [_VOIDP]),
    "pt_worker_handle_message": (C.c_int, [_VOIDP, C.c_char_p, C.POINTER(PtRenderOpts), C.POINTER(C.POINTER(C.c_uint8)),
                                          C.POINTER(C.c_uint64)]),
    "pt_worker_serve_start": (C.c_int, [_VOIDP, C.c_char_p, C.c_uint16, C.c_uint16, C.POINTER(_VOIDP)]),
    "pt_server_ports": (C.c_int, [_VOIDP, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]),
    "pt_server_stats": (C.c_int, [_VOIDP, C.POINTER(PtServerStats), C.c_char_p, C.c_size_t]),
    "pt_server_stop": (None, [_VOIDP]),
    "pt_worker_serve": (C.c_int, [_VOIDP, C.c_char_p, C.c_uint16, C.c_uint16, C.POINTER(C.c_int32)]),
    "pt_frame_create": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_VOIDP)]),
    "pt_frame_destroy": (None, [_VOIDP]),
    "pt_frame_add_render": (C.c_int, [_VOIDP, _VOIDP]),
    "pt_frame_add_render_device": (C.c_int, [_VOIDP, _VOIDP, _VOIDP]),
    "pt_frame_render_count": (C.c_uint64, [_VOIDP]),
    "pt_frame_image": (C.c_int, [_VOIDP, _VOIDP, _VOIDP]),
    "pt_frame_image_device": (C.c_int, [_VOIDP, _VOIDP, _VOIDP, _VOIDP]),
}


class PathTracingError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
        self.message = message


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch wheels bundle their own copy of
    libamdhip64 (same soname as the system one); if this library pulled in the
    system copy and torch later loaded its own, the second runtime would find no
    device.  Loading torch's copy first, by path, makes both resolve to it.
    Without torch installed the system runtime is used."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C pathtracing_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path."
        )
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here = header and library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()


def last_error() -> str:
    buf = C.create_string_buffer(4096)
    lib.pt_last_error(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


def check(status: int) -> None:
    if status != PT_OK:
        raise PathTracingError(status, last_error())


def kernel_fingerprint(path: str = LIB_PATH) -> str:
    """First 16 hex digits of the SHA-256 of the library's device code (its ``.hip_fatbin``
    section): equal for two builds exactly when the kernels are the same machine code,
    whatever the host code around them.  Recorded beside counter measurements, so that a
    reader can tell whether they belong to the kernels it has loaded."""
    import hashlib
    import struct

    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        raise ValueError(f"{path} is not a 64-bit ELF file")
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    sections = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    names = sections[shstrndx]
    for name_off, _type, _flags, _addr, offset, size, *_ in sections:
        end = data.index(b"\0", names[4] + name_off)
        if data[names[4] + name_off:end] == b".hip_fatbin":
            return hashlib.sha256(data[offset:offset + size]).hexdigest()[:16]
    raise ValueError(f"{path} has no .hip_fatbin section")
