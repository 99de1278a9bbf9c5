"""The `.tree` cache end to end on the GPU: files written on a miss, adopted on the
next load (no build), and the adopted structure searched by the kernels gives the
oracle's closest hits and images bit for bit.  Stale and corrupt files are ignored
(silent rebuild) with the same bits.  The reference only names the file
(scene/hierarchy/kd_tree.rs:12-51); loading it is todo!() there (scene/mod.rs:94-96)."""
import os
import time

import numpy as np
import pytest

import pathtracing_amd as P
from pathtracing_amd import synth
from tests import helpers as H
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu

ROUTES = ("thread", "wavefront", "wavefront_split")


def _check_against_oracle(gpu, cpu, rays, what):
    want = cpu.trace(rays, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    for route in ROUTES:
        H.assert_hits_identical(gpu.trace_closest(rays, route=route), want, f"{what} [{route}]")
    on_mesh = (want["hit"] == 1) & (want["tri_index"] != 0xFFFFFFFF)
    assert on_mesh.mean() > 0.02
    sec = H.secondary_rays(want[want["hit"] == 1], rays, seed=3)
    want2 = cpu.trace(sec, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    for route in ROUTES:
        H.assert_hits_identical(gpu.trace_closest(sec, route=route), want2, f"{what} secondary [{route}]")


def _render_pair(gpu, cpu, task, seed=5):
    got = gpu.render(task, P.RenderOpts(seed=seed))
    want, st = cpu.render(task, P.RenderOpts(seed=seed), accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    assert gpu.last_stats.rays == st.rays
    assert np.array_equal(H.bits(got), H.bits(want))
    return got


def test_small_trees_written_adopted_and_searched(oracle, tmp_path):
    """nested.json: cube.obj behind ./cube.tree and gem.obj behind ./gem.tree, both files
    empty as the reference ships cube.tree."""
    d = str(tmp_path / "scene")
    path = H.write_nested_scene(d)
    task = H.load_task()
    task.scene = path
    task.camera.resolution = [64, 48]
    task.config = P.Config(trace_depth=6, iterations=2)
    rays = H.random_rays(60_000, seed=11, origin_box=9.0, target_box=4.0)

    # miss: both files are empty, the structures are built
    scene = P.Scene.load(d, path)
    gpu = P.HipRenderer.init(scene, 0)
    info = gpu.info
    assert (info.trees_offered, info.trees_adopted, info.trees_built) == (0, 0, 2)
    cpu = oracle.scene(scene.desc_pointer)
    _check_against_oracle(gpu, cpu, rays, "built")
    built_image = _render_pair(gpu, cpu, task)
    gpu.close()

    # write-back, then a fresh load: both adopted, nothing built, same bits
    assert scene.cache_trees(d) == 2
    assert os.path.getsize(os.path.join(d, "cube.tree")) > 64 and os.path.getsize(os.path.join(d, "gem.tree")) > 64
    assert scene.cache_trees(d) == 0  # now a hit
    scene2 = P.Scene.load(d, path)
    gpu2 = P.HipRenderer.init(scene2, 0)
    info = gpu2.info
    assert (info.trees_offered, info.trees_adopted, info.trees_built) == (2, 2, 0)
    assert info.build_ms == 0.0
    _check_against_oracle(gpu2, cpu, rays, "adopted")
    assert np.array_equal(H.bits(_render_pair(gpu2, cpu, task)), H.bits(built_image))
    gpu2.close()

    # the scene that took the files into itself adopts them too, without reloading
    gpu3 = P.HipRenderer.init(scene, 0)
    assert gpu3.info.trees_adopted == 2
    gpu3.close()

    # corrupt (truncated) and damaged (a flipped byte in the geometry hash) files: silent rebuild
    blob = open(os.path.join(d, "gem.tree"), "rb").read()
    with open(os.path.join(d, "gem.tree"), "wb") as f:
        f.write(blob[: len(blob) // 2])
    damaged = bytearray(open(os.path.join(d, "cube.tree"), "rb").read())
    damaged[66] ^= 0x40
    with open(os.path.join(d, "cube.tree"), "wb") as f:
        f.write(bytes(damaged))
    scene4 = P.Scene.load(d, path)
    gpu4 = P.HipRenderer.init(scene4, 0)
    info = gpu4.info
    assert info.trees_offered == 2 and info.trees_adopted == 0 and info.trees_built == 2
    _check_against_oracle(gpu4, cpu, rays, "rebuilt after corruption")
    assert np.array_equal(H.bits(_render_pair(gpu4, cpu, task)), H.bits(built_image))
    gpu4.close()
    assert scene4.cache_trees(d) == 2  # and the cache heals

    # stale: the mesh changes under a valid file of the old mesh
    with open(os.path.join(d, "gem.obj")) as f:
        text = f.read().replace("v 0 1.4 0", "v 0 1.9 0")
    with open(os.path.join(d, "gem.obj"), "w") as f:
        f.write(text)
    scene5 = P.Scene.load(d, path)
    gpu5 = P.HipRenderer.init(scene5, 0)
    info = gpu5.info
    assert info.trees_offered == 2 and info.trees_adopted == 1 and info.trees_built == 1
    cpu5 = oracle.scene(scene5.desc_pointer)
    _check_against_oracle(gpu5, cpu5, rays, "rebuilt after the mesh changed")
    _render_pair(gpu5, cpu5, task)
    gpu5.close()
    cpu5.close()
    cpu.close()


def test_dragon_tree_cache(oracle, tmp_path):
    """The 871k-triangle mesh: the first load builds (and the worker writes dragon.tree),
    the second adopts; closest hits of the adopted tree against the oracle's own BVH and
    (a sample) its linear scan, through all three routes; a render bit for bit."""
    out = str(tmp_path)
    task = H.load_task(synth.write_dragon_scene(out, H.SCENE_DATA))
    assert not os.path.exists(os.path.join(out, "dragon.tree"))
    t0 = time.time()
    scene = P.Scene.load(out, task.scene)
    load_s = time.time() - t0
    gpu = P.HipRenderer.init(scene, 0)
    miss = gpu.info
    assert (miss.trees_adopted, miss.trees_built) == (0, 1)
    gpu.close()
    t0 = time.time()
    assert scene.cache_trees(out) == 1
    write_s = time.time() - t0
    scene2 = P.Scene.load(out, task.scene)
    gpu = P.HipRenderer.init(scene2, 0)
    hit = gpu.info
    assert (hit.trees_offered, hit.trees_adopted, hit.trees_built) == (1, 1, 0)
    print(f"dragon scene: load {load_s:.2f} s; pt_scene_create {miss.setup_ms / 1e3:.2f} s on a miss (build "
          f"{miss.build_ms / 1e3:.2f} s), {hit.setup_ms / 1e3:.2f} s on a hit; writing the file {write_s:.2f} s")
    assert hit.setup_ms < miss.setup_ms

    cpu = oracle.scene(scene2.desc_pointer)
    rng = np.random.default_rng(5)
    n = 200_000
    px = np.stack([rng.integers(0, 1920, n), rng.integers(0, 1080, n)], axis=1).astype(np.uint32)
    rays = P.generate_rays(task.camera, 0, px, rng.integers(0, 64, n).astype(np.uint64))
    want = cpu.trace(rays, accel=ob.ACCEL_BVH, math=ob.MATH_PORTABLE)
    on_mesh = (want["hit"] == 1) & (want["tri_index"] != 0xFFFFFFFF)
    sec = H.secondary_rays(want[on_mesh], rays[on_mesh], seed=6)
    want2 = cpu.trace(sec, accel=ob.ACCEL_BVH, math=ob.MATH_PORTABLE)
    for route in ROUTES:
        H.assert_hits_identical(gpu.trace_closest(rays, route=route), want, f"adopted dragon.tree, primary [{route}]")
        H.assert_hits_identical(gpu.trace_closest(sec, route=route), want2, f"adopted dragon.tree, secondary [{route}]")
    sel = np.concatenate([rays[on_mesh][:200], sec[:200]])
    lin = cpu.trace(sel, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    H.assert_hits_identical(gpu.trace_closest(sel, route="wavefront"), lin, "adopted dragon.tree vs the oracle's linear scan")
    task.camera.resolution = [160, 90]
    task.config.iterations = 2
    got = gpu.render(task, P.RenderOpts(seed=1))
    ref, st = cpu.render(task, P.RenderOpts(seed=1), accel=ob.ACCEL_BVH, math=ob.MATH_PORTABLE)
    assert gpu.last_stats.rays == st.rays and np.array_equal(H.bits(got), H.bits(ref))
    gpu.close()
    cpu.close()
