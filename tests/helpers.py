"""Shared helpers for the parity tests."""
from __future__ import annotations

import json
import math
import os
import shutil

import numpy as np

import pathtracing_amd as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE_DATA = os.path.join(ROOT, "scene_data")
F32_MAX = np.float32(3.4028234663852886e38)


def load_task(path=None) -> P.RenderTask:
    path = path or os.path.join(SCENE_DATA, "render_task.json")
    with open(path) as f:
        return P.RenderTask.from_json(f.read())


def random_rays(n: int, seed: int, origin_box=6.0, target_box=3.0, t_min=0.0) -> np.ndarray:
    """Rays from random origins toward random targets (unit directions)."""
    rng = np.random.default_rng(seed)
    src = rng.uniform(-origin_box, origin_box, size=(n, 3)).astype(np.float32)
    dst = rng.uniform(-target_box, target_box, size=(n, 3)).astype(np.float32)
    d = (dst - src).astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = src
    rays[:, 3:6] = d.astype(np.float32)
    rays[:, 6] = t_min
    rays[:, 7] = F32_MAX
    return rays


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_hits_identical(got: np.ndarray, want: np.ndarray, what: str = "") -> None:
    """hit flag, leaf node, triangle index, material, t / point / normal / uv bit for bit.
    Records of misses carry no defined t or point, so only the flag is compared there."""
    assert got.shape == want.shape
    assert np.array_equal(got["hit"], want["hit"]), f"{what}: hit flags differ at {np.nonzero(got['hit'] != want['hit'])[0][:8]}"
    m = want["hit"] == 1
    if not m.any():
        return
    for name in ("hit_inside", "node_index", "tri_index", "material_id"):
        bad = np.nonzero(got[name][m] != want[name][m])[0]
        assert bad.size == 0, f"{what}: {name} differs for {bad.size} hits, first {bad[:8]}"
    for name in ("t", "point", "normal", "uv"):
        g, w = bits(got[name][m]), bits(want[name][m])
        bad = np.nonzero((g != w).reshape(g.shape[0], -1).any(axis=1))[0]
        assert bad.size == 0, f"{what}: {name} differs in bits for {bad.size} hits, first {bad[:8]}"


ROUTES = ("thread", "wavefront", "wavefront_split")


def assert_all_routes(gpu, rays: np.ndarray, want: np.ndarray, what: str = "") -> None:
    """The closest hits of `rays` through every route of pt_trace_closest: one thread per
    ray with its searches in place, and the kernels a render uses (queue regions, listed
    searches, search_kernel, then the fused walk + bounce kernel or the separate walk kernel)."""
    for route in ROUTES:
        assert_hits_identical(gpu.trace_closest(rays, route=route), want, f"{what} [{route}]")


def secondary_rays(hits: np.ndarray, rays: np.ndarray, seed: int) -> np.ndarray:
    """Bounce rays leaving the hit points of `hits` in random hemisphere directions,
    built like work_group.rs:42-46 (origin = point + dir * 1e-4, min = 1e-4)."""
    m = hits["hit"] == 1
    pts = hits["point"][m]
    nrm = hits["normal"][m]
    rng = np.random.default_rng(seed)
    d = rng.normal(size=pts.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    flip = (d * nrm).sum(1) < 0
    d[flip] *= -1
    d = d.astype(np.float32)
    out = np.zeros((pts.shape[0], 8), dtype=np.float32)
    out[:, 0:3] = pts + d * np.float32(1e-4)
    out[:, 3:6] = d
    out[:, 6] = 1e-4
    out[:, 7] = F32_MAX
    return out


def rmse(a: np.ndarray, b: np.ndarray) -> float:
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def _translate(x, y, z):
    return [[1.0, 0.0, 0.0, x], [0.0, 1.0, 0.0, y], [0.0, 0.0, 1.0, z], [0.0, 0.0, 0.0, 1.0]]


def _trs(angle_y, scale, pos):
    c, s = math.cos(angle_y), math.sin(angle_y)
    sx, sy, sz = scale
    return [[c * sx, 0.0, s * sz, pos[0]], [0.0, sy, 0.0, pos[1]], [-s * sx, 0.0, c * sz, pos[2]], [0.0, 0.0, 0.0, 1.0]]


GEM_OBJ = """# octahedron with vertex normals and texture coordinates, one quad-free model per half
o top
v 1 0 0
v -1 0 0
v 0 1.4 0
v 0 -1.4 0
v 0 0 1
v 0 0 -1
vn 1 0 0
vn -1 0 0
vn 0 1 0
vn 0 -1 0
vn 0 0 1
vn 0 0 -1
vt 0 0
vt 1 0
vt 0.5 1
f 1/1/1 3/3/3 5/2/5
f 5/1/5 3/3/3 2/2/2
f 2/1/2 3/3/3 6/2/6
f 6/1/6 3/3/3 1/2/1
o bottom
f 5 4 1
f 2 4 5
f 6 4 2
f 1 4 6
"""


def write_nested_scene(directory: str) -> str:
    """A copy of the scene fixtures plus `nested.json`: collections and transforms
    nested three deep, six mesh nodes over two meshes (one of them twice behind
    kd_tree nodes, one directly in world space), spheres between them, a glass and
    a mirror material.  Returns the scene path relative to `directory`."""
    if os.path.isdir(directory):
        shutil.rmtree(directory)
    shutil.copytree(SCENE_DATA, directory)
    with open(os.path.join(directory, "gem.obj"), "w") as f:
        f.write(GEM_OBJ)
    with open(os.path.join(directory, "gem.tree"), "w") as f:
        pass  # empty like the reference's cube.tree: the tree is rebuilt
    with open(os.path.join(directory, "materials", "glass.json"), "w") as f:
        json.dump({"type": "base", "refractive": 0.9, "refraction": 1.5, "reflective": 0.05}, f)

    def mesh(path, mat, tree=None):
        node = {"type": "mesh", "path": path, "material": f"./materials/{mat}.json"}
        return {"type": "kd_tree", "path": tree, "child": node} if tree else node

    def plane(point, normal, tangent, bitangent, mat):
        return {"type": "plane", "point": point, "normal": normal, "tangent": tangent, "bitangent": bitangent,
                "material": f"./materials/{mat}.json"}

    scene = {"type": "node_collection", "children": [
        {"type": "sphere", "center": [0.0, 6.0, 2.0], "radius": 2.0, "material": "./materials/mat0.json"},
        {"type": "transform", "matrix": _trs(0.6, (1.0, 0.5, 1.5), (-2.0, -1.0, 0.0)), "child": {
            "type": "node_collection", "children": [
                mesh("./cube.obj", "mat3", "./cube.tree"),
                {"type": "sphere", "center": [0.0, 3.0, 0.0], "radius": 1.0, "material": "./materials/glass.json"},
                {"type": "transform", "matrix": _trs(-1.1, (0.5, 0.5, 0.5), (3.0, 3.0, 1.0)), "child": {
                    "type": "node_collection", "children": [
                        mesh("./gem.obj", "mat5"),
                        {"type": "transform", "matrix": _translate(0.0, 3.5, 0.0), "child": mesh("./cube.obj", "mat1")},
                    ]}},
            ]}},
        mesh("./gem.obj", "glass", "./gem.tree"),
        plane([0.0, -3.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0], "mat2"),
        {"type": "transform", "matrix": _trs(0.3, (0.7, 0.7, 0.7), (2.5, -1.5, 1.0)), "child": mesh("./cube.obj", "mat4")},
        {"type": "node_collection", "children": []},
        {"type": "transform", "matrix": _trs(2.0, (1.0, 2.0, 1.0), (2.0, 1.5, -1.5)), "child": mesh("./gem.obj", "mat1")},
        plane([0.0, 0.0, -5.0], [0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], "mat3"),
    ]}
    with open(os.path.join(directory, "nested.json"), "w") as f:
        json.dump(scene, f, indent=1)
    return "./nested.json"
