"""Shared helpers for the parity tests."""
from __future__ import annotations

import os

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
    for name in ("hit_inside", "node_index", "tri_index", "material_id"):
        bad = np.nonzero(got[name][m] != want[name][m])[0]
        assert bad.size == 0, f"{what}: {name} differs for {bad.size} hits, first {bad[:8]}"
    for name in ("t", "point", "normal", "uv"):
        g, w = bits(got[name][m]), bits(want[name][m])
        bad = np.nonzero((g != w).reshape(g.shape[0], -1).any(axis=1))[0]
        assert bad.size == 0, f"{what}: {name} differs in bits for {bad.size} hits, first {bad[:8]}"


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
