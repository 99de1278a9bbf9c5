"""Writes the golden vectors under tests/golden/ (run from the repository root:
``python tests/golden/make_golden.py``).

The reference is Rust and cannot be run in this environment, and it ships no test
vectors of its own, so these vectors come from the CPU oracle (the restatement of
the reference's algorithm that tests/test_oracle_kat.py pins with hand-derived
known answers), in its literal mode: linear scan of every triangle, portable
transcendentals.  Each file holds inputs and expected outputs only.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import pathtracing_amd as P  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests import oracle_binding as ob  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SCENES = ["scene0", "simple_scene", "scene", "nested"]
N_RAYS = 3000
RENDER = dict(resolution=[40, 28], iterations=2, trace_depth=8, seed=5)


def scene_root(name: str, tmp: str) -> str:
    if name == "nested":
        root = os.path.join(tmp, "nested")
        H.write_nested_scene(root)
        return root
    return H.SCENE_DATA


def main() -> None:
    oracle = ob.load()
    # random stream: draws of a few (seed, stream, counter) triples
    rng = np.random.default_rng(2024)
    seeds = rng.integers(0, 1 << 63, 256, dtype=np.uint64)
    streams = rng.integers(0, 1 << 63, 256, dtype=np.uint64)
    draw_index = rng.integers(0, 64, 256).astype(np.uint32)  # 0-based: draw i uses counter i + 1
    u32 = np.array([oracle.lib.oracle_rng_u32(int(a), int(b), int(c)) for a, b, c in zip(seeds, streams, draw_index)],
                   dtype=np.uint32)
    rng_range = np.array([oracle.lib.oracle_rng_range(int(a), int(b), int(c), -0.5, 0.5)
                          for a, b, c in zip(seeds, streams, draw_index)], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "rng.npz"), seed=seeds, stream=streams, draw_index=draw_index, u32=u32,
                        range_half=rng_range)

    # camera rays for the three lens shapes
    task = H.load_task()
    cams = []
    for shape, size in (("point", 0.0), ("circle", 0.35), ("square", 0.2)):
        task.camera.bokeh_shape, task.camera.bokeh_size = shape, size
        px = np.stack([rng.integers(0, 1024, 2000), rng.integers(0, 1024, 2000)], axis=1).astype(np.uint32)
        smp = rng.integers(0, 1 << 20, 2000).astype(np.uint64)
        rays = oracle.generate_rays(task.camera, 7, px, smp, math=ob.MATH_PORTABLE)
        cams.append((shape, size, px, smp, rays))
    np.savez_compressed(os.path.join(OUT, "camera.npz"), seed=np.uint64(7),
                        **{f"{s}_size": np.float32(z) for s, z, _, _, _ in cams},
                        **{f"{s}_pixels": p for s, _, p, _, _ in cams},
                        **{f"{s}_samples": m for s, _, _, m, _ in cams},
                        **{f"{s}_rays": r for s, _, _, _, r in cams})

    with tempfile.TemporaryDirectory() as tmp:
        for name in SCENES:
            root = scene_root(name, tmp)
            scene = P.Scene.load(root, f"./{name}.json")
            osc = oracle.scene(scene.desc_pointer)
            rays = H.random_rays(N_RAYS, seed=31, origin_box=8.0, target_box=3.5)
            hits = osc.trace(rays, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
            sec = H.secondary_rays(hits, rays, seed=32)
            hits2 = osc.trace(sec, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
            m = hits["hit"] == 1
            sid = np.arange(int(m.sum()), dtype=np.uint64) * 977 + 5
            kind, a, d = osc.shade(rays[m, 3:6], hits[m], sid, seed=99, math=ob.MATH_PORTABLE)
            t = H.load_task()
            t.camera.resolution = list(RENDER["resolution"])
            t.config = P.Config(trace_depth=RENDER["trace_depth"], iterations=RENDER["iterations"])
            image, stats = osc.render(t, P.RenderOpts(seed=RENDER["seed"]), accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
            np.savez_compressed(os.path.join(OUT, f"{name}.npz"), rays=rays, hits=hits, secondary_rays=sec,
                                secondary_hits=hits2, shade_stream=sid, shade_seed=np.uint64(99), shade_kind=kind,
                                shade_a=a, shade_dir=d, render_resolution=np.array(RENDER["resolution"]),
                                render_iterations=np.int64(RENDER["iterations"]),
                                render_trace_depth=np.int64(RENDER["trace_depth"]), render_seed=np.uint64(RENDER["seed"]),
                                render_image=image, render_rays=np.uint64(stats.rays))
            osc.close()
            print(name, "hit rate", float(m.mean()), "secondary", float((hits2["hit"] == 1).mean()), "mean", float(image.mean()))


if __name__ == "__main__":
    main()
