"""Deterministic synthetic scenes for the benchmark configurations.

The reference's ``scene_data/dragon.obj`` (~871k triangles) is not distributed
with it, and no scene file references it.  ``write_dragon_scene`` generates a
stand-in with the same triangle count: a closed, bumpy (2,3) torus-knot tube,
written as a Wavefront OBJ, inside a box of textured planes lit by an emissive
ceiling and a light sphere, plus the render task that looks at it.

The model is ~23 units across on purpose: the reference's triangle test rejects
``|det| < 1e-4`` (``triangle/mod.rs:140``) and ``det`` is about twice the
triangle's area times the cosine of incidence, so a model this dense has to be
large for its triangles to be visible at all but grazing angles.
"""
from __future__ import annotations

import json
import math
import os
import shutil

import numpy as np

DRAGON_NU = 1320  # segments along the knot
DRAGON_NV = 330   # segments around the tube -> 2 * NU * NV = 871,200 triangles


def knot_mesh(nu: int = DRAGON_NU, nv: int = DRAGON_NV, scale: float = 2.0):
    """Vertices float32 [nu*nv, 3] and triangles int32 [2*nu*nv, 3] of the knot tube."""
    u = (np.arange(nu, dtype=np.float64) / nu) * 2.0 * np.pi
    v = (np.arange(nv, dtype=np.float64) / nv) * 2.0 * np.pi
    p, q, R, a = 2.0, 3.0, 3.3, 1.4
    cu, su = np.cos(p * u), np.sin(p * u)
    cq, sq = np.cos(q * u), np.sin(q * u)
    centre = np.stack([(R + a * cq) * cu, (R + a * cq) * su, a * sq], axis=1)
    # derivative of the centre line
    d = np.stack(
        [-a * q * sq * cu - (R + a * cq) * p * su, -a * q * sq * su + (R + a * cq) * p * cu, a * q * cq], axis=1
    )
    t = d / np.linalg.norm(d, axis=1, keepdims=True)
    # outward direction of the carrying torus, made orthogonal to the tangent
    n0 = np.stack([cq * cu, cq * su, sq], axis=1)
    n = n0 - (n0 * t).sum(1, keepdims=True) * t
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    b = np.cross(t, n)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    radius = 0.75 * (1.0 + 0.22 * np.sin(9.0 * uu) * np.cos(4.0 * vv) + 0.10 * np.sin(31.0 * uu + 5.0 * vv)
                     + 0.05 * np.cos(57.0 * uu - 11.0 * vv))
    pos = centre[:, None, :] + radius[..., None] * (np.cos(vv)[..., None] * n[:, None, :] + np.sin(vv)[..., None] * b[:, None, :])
    verts = (pos.reshape(-1, 3) * scale).astype(np.float32)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    i1, j1 = (i + 1) % nu, (j + 1) % nv
    a00 = (i * nv + j).ravel()
    a10 = (i1 * nv + j).ravel()
    a11 = (i1 * nv + j1).ravel()
    a01 = (i * nv + j1).ravel()
    tris = np.empty((2 * nu * nv, 3), dtype=np.int32)
    tris[0::2] = np.stack([a00, a10, a11], axis=1)
    tris[1::2] = np.stack([a00, a11, a01], axis=1)
    return verts, tris


def write_obj(path: str, verts: np.ndarray, tris: np.ndarray) -> None:
    with open(path, "wb") as f:
        f.write(b"# synthetic stand-in for dragon.obj (torus-knot tube)\no knot\n")
        np.savetxt(f, verts, fmt="v %.9g %.9g %.9g")
        np.savetxt(f, tris + 1, fmt="f %d %d %d")


def _plane(point, normal, tangent, bitangent, material):
    return {"type": "plane", "point": point, "normal": normal, "tangent": tangent, "bitangent": bitangent,
            "material": material}


def dragon_scene_json(half: float = 16.0, use_kd_tree: bool = True) -> dict:
    mesh = {"type": "mesh", "path": "./dragon.obj", "material": "./materials/mat4.json"}
    if use_kd_tree:
        mesh = {"type": "kd_tree", "path": "./dragon.tree", "child": mesh}
    return {
        "type": "node_collection",
        "children": [
            {"type": "sphere", "center": [0.0, half, half], "radius": 4.0, "material": "./materials/mat0.json"},
            {
                "type": "transform",
                "matrix": [[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, -2.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]],
                "child": mesh,
            },
            _plane([0.0, -half, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0], "./materials/mat2.json"),
            _plane([0.0, half, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0], "./materials/mat0.json"),
            _plane([0.0, 0.0, -half], [0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], "./materials/mat3.json"),
            _plane([-half, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0], "./materials/mat2.json"),
            _plane([half, 0.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], "./materials/mat5.json"),
        ],
    }


def dragon_render_task(width: int = 1920, height: int = 1080, iterations: int = 64, trace_depth: int = 8) -> dict:
    return {
        "scene": "./dragon_scene.json",
        "config": {"trace_depth": trace_depth, "iterations": iterations},
        "camera": {
            "resolution": [width, height],
            "rotation": [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]],
            "position": [0.0, 0.0, 70.0],
            "field_of_view": 0.523,
            "near_plane": 0.0,
            "focal_length": 70.0,
            "bokeh_shape": "circle",
            "bokeh_size": 0.0,
        },
    }


def write_dragon_scene(out_dir: str, fixtures_dir: str, nu: int = DRAGON_NU, nv: int = DRAGON_NV) -> str:
    """Writes dragon.obj, dragon_scene.json, dragon_task.json, the materials and the
    textures they use into ``out_dir``.  Returns the path of the task file."""
    os.makedirs(os.path.join(out_dir, "materials"), exist_ok=True)
    for name in os.listdir(os.path.join(fixtures_dir, "materials")):
        shutil.copy(os.path.join(fixtures_dir, "materials", name), os.path.join(out_dir, "materials", name))
    for name in ("test_texture_0.png", "test_texture_1.png"):
        shutil.copy(os.path.join(fixtures_dir, name), os.path.join(out_dir, name))
    obj = os.path.join(out_dir, "dragon.obj")
    marker = os.path.join(out_dir, f".dragon_{nu}x{nv}")
    if not (os.path.exists(obj) and os.path.exists(marker)):
        verts, tris = knot_mesh(nu, nv)
        write_obj(obj, verts, tris)
        open(marker, "w").close()
    with open(os.path.join(out_dir, "dragon_scene.json"), "w") as f:
        json.dump(dragon_scene_json(), f, indent=1)
    task = os.path.join(out_dir, "dragon_task.json")
    with open(task, "w") as f:
        json.dump(dragon_render_task(), f, indent=1)
    return task


def instanced_dragon_scene_json(count: int, seed: int = 0, half: float = 40.0, spacing: float = 26.0) -> dict:
    """`count` transform -> kd_tree -> mesh instances of dragon.obj on a jittered grid
    (rotation about y, uniform scale 1.0..1.3, seeded) inside a larger box: the
    instanced workload, with the scene graph itself as the list of instances."""
    import math
    import random

    rnd = random.Random(seed)
    side = max(1, math.ceil(math.sqrt(count)))
    children = [{"type": "sphere", "center": [0.0, half, half], "radius": 8.0, "material": "./materials/mat0.json"}]
    for i in range(count):
        gx, gz = i % side, i // side
        x = (gx - (side - 1) / 2) * spacing + rnd.uniform(-2.0, 2.0)
        z = (gz - (side - 1) / 2) * spacing + rnd.uniform(-2.0, 2.0)
        a, sc = rnd.uniform(0.0, 2.0 * math.pi), rnd.uniform(1.0, 1.3)
        c, s_ = math.cos(a) * sc, math.sin(a) * sc
        matrix = [[c, 0.0, s_, x], [0.0, sc, 0.0, -2.0], [-s_, 0.0, c, z], [0.0, 0.0, 0.0, 1.0]]
        mesh = {"type": "mesh", "path": "./dragon.obj", "material": "./materials/mat4.json" if i % 2 else "./materials/mat5.json"}
        children.append({"type": "transform", "matrix": matrix,
                         "child": {"type": "kd_tree", "path": "./dragon.tree", "child": mesh}})
    children += [
        _plane([0.0, -half, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0], "./materials/mat2.json"),
        _plane([0.0, half, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0], "./materials/mat0.json"),
        _plane([0.0, 0.0, -half], [0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], "./materials/mat3.json"),
    ]
    return {"type": "node_collection", "children": children}


def flattened_scene_json(half: float = 110.0) -> dict:
    """One identity transform -> mesh of dragon.obj (its triangles are swapped for
    the flattened copies after loading) between a floor, a lit ceiling and a back
    wall `half` away: the scene graph of the flattened-instances workload."""
    identity = [[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]]
    return {
        "type": "node_collection",
        "children": [
            {"type": "sphere", "center": [0.0, half, half], "radius": 20.0, "material": "./materials/mat0.json"},
            {"type": "transform", "matrix": identity,
             "child": {"type": "mesh", "path": "./dragon.obj", "material": "./materials/mat4.json"}},
            _plane([0.0, -half, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0], "./materials/mat2.json"),
            _plane([0.0, half, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0], "./materials/mat0.json"),
            _plane([0.0, 0.0, -half], [0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], "./materials/mat3.json"),
        ],
    }


def flattened_copies(one: np.ndarray, copies: int, seed: int = 0, spacing: float = 26.0, progress=None) -> np.ndarray:
    """`copies` rotated (about y), scaled (1.0..1.3) and shifted copies of the
    triangles `one` ([n, 9] float32 positions) on a jittered cubic grid centred on
    the origin, as one array [copies * n, 9] of world-space triangles."""
    import random

    rnd = random.Random(seed)
    side = 1
    while side ** 3 < copies:
        side += 1
    n1 = one.shape[0]
    verts = np.ascontiguousarray(one, dtype=np.float32).reshape(n1 * 3, 3)
    out = np.empty((copies * n1, 9), dtype=np.float32)
    for i in range(copies):
        g = (i % side, (i // side) % side, i // (side * side))
        shift = np.array([(g[k] - (side - 1) / 2) * spacing + rnd.uniform(-2.0, 2.0) for k in range(3)], dtype=np.float32)
        a, sc = rnd.uniform(0.0, 2.0 * math.pi), rnd.uniform(1.0, 1.3)
        c, s_ = math.cos(a) * sc, math.sin(a) * sc
        rot = np.array([[c, 0.0, s_], [0.0, sc, 0.0], [-s_, 0.0, c]], dtype=np.float32)
        out[i * n1:(i + 1) * n1] = (verts @ rot.T + shift).reshape(n1, 9)
        if progress is not None and (i + 1) % 16 == 0:
            progress(f"copies written: {i + 1}/{copies}")
    return out
