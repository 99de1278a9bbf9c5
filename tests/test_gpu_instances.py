"""Instance runs: many transform -> (kd_tree ->) mesh children of the root collection,
reached through a tree over their world-space boxes instead of one after the other.
The result has to be what the reference's loop over the children returns
(node_collection.rs:43-55): smallest t, earliest child among equals."""
import json
import math
import os
import shutil

import numpy as np
import pytest

import pathtracing_amd as P
from tests import helpers as H
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu

RUN_MIN = "PATHTRACING_AMD_INSTANCE_RUN_MIN"


def _instance(i, matrix, kd=False):
    path = "./gem.obj" if i % 3 == 0 else "./cube.obj"
    node = {"type": "mesh", "path": path, "material": f"./materials/mat{1 + i % 5}.json"}
    if kd:
        node = {"type": "kd_tree", "path": path.replace(".obj", ".tree"), "child": node}
    return {"type": "transform", "matrix": matrix, "child": node}


def _plane(point, normal, tangent, bitangent, mat):
    return {"type": "plane", "point": point, "normal": normal, "tangent": tangent, "bitangent": bitangent,
            "material": f"./materials/{mat}.json"}


def write_run_scene(directory, crowded=False):
    """crowded=False: a light, four instances no run can hold, a plane, a short run of
    five (walked child by child), a sphere, a run of 36 on a jittered grid with two exact
    duplicates, two untransformed meshes and overlapping neighbours, a nested collection, a back wall.
    crowded=True: 40 instances stacked on nearly the same spot and filling the frame, so
    every camera ray meets all forty boxes (five times the job slots set aside per ray: the
    regions' job lists fill up and the remaining searches happen while walking)."""
    if os.path.isdir(directory):
        shutil.rmtree(directory)
    shutil.copytree(H.SCENE_DATA, directory)
    with open(os.path.join(directory, "gem.obj"), "w") as f:
        f.write(H.GEM_OBJ)
    open(os.path.join(directory, "gem.tree"), "w").close()
    rnd = np.random.default_rng(12)
    children = [{"type": "sphere", "center": [0.0, 9.0, 2.0], "radius": 3.0, "material": "./materials/mat0.json"}]
    if crowded:
        for i in range(40):
            m = H._trs(0.05 * i, (2.2 + 0.02 * i,) * 3, (0.02 * i, -1.0, 0.01 * i))
            children.append(_instance(i, m, kd=i % 2 == 0))
    else:
        # four children outside any run: unequal scale factors (under those the reference's ray leaves
        # the line of the world-space ray, ray.rs:21-37) and a matrix without an inverse (the
        # reference's rays turn non-finite in there and hit nothing)
        for i in range(3):
            children.append(_instance(i, H._trs(0.4 * i, (0.5, 0.3 + 0.3 * i, 0.7), (-6.0 + 1.5 * i, 2.5, -2.0))))
        children.append(_instance(4, [[1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]]))
        children.append(_plane([0.0, -3.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0], "mat2"))
        # a short run (five children, one mirrored), then the long one
        for i in range(5):
            sc = 0.3 + 0.05 * i
            m = H._trs(0.9 * i, (sc, sc, -sc if i == 2 else sc), (5.5, -2.0 + 1.2 * i, -3.0 + 0.3 * i))
            children.append(_instance(i + 1, m))
        children.append({"type": "sphere", "center": [-6.0, -1.0, 3.0], "radius": 1.0, "material": "./materials/mat1.json"})
        for i in range(32):
            x, z = (i % 8 - 3.5) * 1.7, (i // 8 - 1.5) * 1.9
            sc = 0.35 + 0.45 * rnd.random()
            m = H._trs(rnd.uniform(0, 2 * math.pi), (sc, sc, sc),
                       (x + rnd.uniform(-0.4, 0.4), -1.5 + rnd.uniform(0, 1.5), z + rnd.uniform(-0.4, 0.4)))
            children.append(_instance(i, m, kd=i % 4 == 1))
            if i == 9:  # meshes standing in the collection without a transform belong to the run as well
                children.append({"type": "mesh", "path": "./gem.obj", "material": "./materials/mat4.json"})
                children.append({"type": "kd_tree", "path": "./cube.tree",
                                 "child": {"type": "mesh", "path": "./cube.obj", "material": "./materials/mat5.json"}})
            if i == 5 or i == 20:  # an exact duplicate right behind it, another material: the first one wins every tie
                dup = _instance(i, m)
                dup["child"]["material"] = "./materials/mat0.json" if i == 5 else "./materials/mat2.json"
                children.append(dup)
        children.append({"type": "transform", "matrix": H._trs(0.3, (0.6, 0.6, 0.6), (0.0, 3.0, 1.0)), "child": {
            "type": "node_collection", "children": [
                {"type": "mesh", "path": "./gem.obj", "material": "./materials/mat4.json"},
                {"type": "sphere", "center": [2.5, 0.0, 0.0], "radius": 0.8, "material": "./materials/mat1.json"}]}})
    children.append(_plane([0.0, 0.0, -7.0], [0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], "mat3"))
    with open(os.path.join(directory, "run.json"), "w") as f:
        json.dump({"type": "node_collection", "children": children}, f)
    return "./run.json"


def _rays(n, seed):
    rays = H.random_rays(n, seed=seed, origin_box=9.0, target_box=6.0)
    # a share of rays that run along coordinate planes and axes (boxes' boundary planes, zero components)
    rays[: n // 20, 4] = 0.0
    rays[: n // 40, 5] = 0.0
    d = rays[: n // 20, 3:6]
    norm = np.linalg.norm(d, axis=1, keepdims=True)
    rays[: n // 20, 3:6] = np.where(norm > 0, d / np.maximum(norm, 1e-30), [[1.0, 0.0, 0.0]]).astype(np.float32)
    return rays


@pytest.mark.parametrize("crowded", [False, True])
def test_instance_run_against_the_oracle(oracle, tmp_path, monkeypatch, crowded):
    root = str(tmp_path / "run")
    path = write_run_scene(root, crowded)
    scene = P.Scene.load(root, path)
    monkeypatch.delenv(RUN_MIN, raising=False)
    gpu = P.HipRenderer.init(scene, 0)
    cpu = oracle.scene(scene.desc_pointer)
    # closest hits: every candidate child searched by the lane that walks the scene graph
    rays = _rays(120_000, 17)
    want = cpu.trace(rays, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    # ... and the same rays through the render's kernels: searches listed through the box tree
    # and run ahead by search_kernel, the walk following each ray's chain of jobs
    H.assert_all_routes(gpu, rays, want, "instance run, closest hits")
    on_mesh = (want["hit"] == 1) & (want["tri_index"] != 0xFFFFFFFF)
    assert np.unique(want["node_index"][on_mesh]).size >= (30 if not crowded else 10)
    sec = H.secondary_rays(want, rays, seed=5)[:60_000]
    H.assert_all_routes(gpu, sec, cpu.trace(sec, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE), "instance run, secondary rays")
    # renders: the searches listed through the box tree and run ahead of the walk
    task = H.load_task()
    task.camera.resolution = [96, 72]
    task.config = P.Config(trace_depth=6, iterations=3)
    img = gpu.render(task, P.RenderOpts(seed=4))
    ref, st = cpu.render(task, P.RenderOpts(seed=4), accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    assert gpu.last_stats.rays == st.rays
    assert np.array_equal(H.bits(img), H.bits(ref))
    # tiny batches (regions of a few rays) and the linear-scan mode change nothing
    assert np.array_equal(H.bits(gpu.render(task, P.RenderOpts(seed=4, paths_per_batch=2048))), H.bits(ref))
    assert np.array_equal(H.bits(gpu.render(task, P.RenderOpts(seed=4, brute_force=True))), H.bits(ref))
    gpu.close()
    # the same scene with every child walked one after the other
    monkeypatch.setenv(RUN_MIN, "0")
    plain = P.HipRenderer.init(scene, 0)
    assert np.array_equal(H.bits(plain.render(task, P.RenderOpts(seed=4))), H.bits(ref))
    H.assert_all_routes(plain, rays, want, "children one after the other")
    plain.close()
    cpu.close()


def test_duplicate_children_first_one_wins(oracle, tmp_path, monkeypatch):
    """Two identical instances: every hit is an exact tie, and the reference keeps the first child."""
    root = str(tmp_path / "dup")
    shutil.copytree(H.SCENE_DATA, root)
    m = H._trs(0.7, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0))
    children = []
    for i in range(6):
        children.append({"type": "transform", "matrix": m if i in (2, 4) else H._translate(3.0 * i - 8.0, 0.0, -3.0),
                         "child": {"type": "mesh", "path": "./cube.obj", "material": f"./materials/mat{i % 5 + 1}.json"}})
    with open(os.path.join(root, "dup.json"), "w") as f:
        json.dump({"type": "node_collection", "children": children}, f)
    scene = P.Scene.load(root, "./dup.json")
    monkeypatch.delenv(RUN_MIN, raising=False)
    gpu = P.HipRenderer.init(scene, 0)
    cpu = oracle.scene(scene.desc_pointer)
    rays = H.random_rays(40_000, seed=3, origin_box=7.0, target_box=1.5)
    want = cpu.trace(rays, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    H.assert_all_routes(gpu, rays, want, "duplicates")
    hit = want["hit"] == 1
    mesh_nodes = [i for i, n in enumerate(scene.desc.nodes[: scene.desc.node_count]) if n.type == 2]
    assert (want["node_index"][hit] == mesh_nodes[2]).any() and not (want["node_index"][hit] == mesh_nodes[4]).any()
    task = H.load_task()
    task.camera.resolution = [64, 48]
    task.config = P.Config(trace_depth=4, iterations=2)
    ref, _ = cpu.render(task, P.RenderOpts(seed=2), accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    assert np.array_equal(H.bits(gpu.render(task, P.RenderOpts(seed=2))), H.bits(ref))
    gpu.close()
    cpu.close()
