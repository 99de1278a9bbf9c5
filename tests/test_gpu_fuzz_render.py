"""Random materials, cameras and render options against the oracle, images bit for
bit: base materials with every mix of reflective / refractive / emissive shares,
plain and textured colours in both uv modes, PBR materials over the whole roughness /
metallic square, cameras with every bokeh shape, rotations, near planes and
resolutions that are no multiple of anything, tiles, sample offsets and batch sizes.
PATHTRACING_AMD_FUZZ_SEEDS sets how many cases are tried (default 40),
PATHTRACING_AMD_FUZZ_FIRST the first seed (default 0)."""
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

N_MATERIALS = 7


def random_material(rnd):
    u = lambda lo, hi: float(rnd.uniform(lo, hi))  # noqa: E731
    if rnd.random() < 0.35:
        return {"type": "pbr", "albedo": [u(0, 1), u(0, 1), u(0, 1)],
                "roughness": [0.0, 1.0, u(0, 1), u(0.01, 0.2)][int(rnd.integers(0, 4))],
                "metallic": [0.0, 1.0, u(0, 1)][int(rnd.integers(0, 3))]}
    m = {"type": "base"}
    r = rnd.random()
    if r < 0.4:
        m["color"] = {"type": "color", "color": [u(0, 1), u(0, 1), u(0, 1)]}
    elif r < 0.8:
        m["color"] = {"type": "texture", "path": f"./test_texture_{int(rnd.integers(0, 2))}.png",
                      "uv_mode": "repeat" if rnd.random() < 0.5 else "clamp"}
    # shares of the branches of base.rs; they need not add up to one, and may be absent
    if rnd.random() < 0.5:
        m["reflective"] = [u(0, 1), 1.0, 0.0][int(rnd.integers(0, 3))]
    if rnd.random() < 0.4:
        m["refractive"] = u(0, 1)
        m["refraction"] = [1.0, 1.33, 1.5, u(0.5, 2.5)][int(rnd.integers(0, 4))]
    if rnd.random() < 0.25:
        m["emissive"] = u(0, 1)
        m["emission"] = [u(0, 6), u(0, 6), u(0, 6)]
    return m


def random_scene(rnd):
    mat = lambda: f"./materials/r{int(rnd.integers(0, N_MATERIALS))}.json"  # noqa: E731
    u = lambda lo, hi: float(rnd.uniform(lo, hi))  # noqa: E731
    children = [{"type": "sphere", "center": [0.0, 7.0, 3.0], "radius": 3.0, "material": "./materials/mat0.json"}]
    for _ in range(int(rnd.integers(1, 4))):
        children.append({"type": "sphere", "center": [u(-3, 3), u(-2, 2), u(-3, 2)], "radius": u(0.5, 1.6), "material": mat()})
    children.append({"type": "transform", "matrix": H._trs(u(0, 6), (u(0.5, 1.0),) * 3, (u(-3, 3), u(-2, 0), u(-3, 1))),
                     "child": {"type": "kd_tree", "path": "./gem.tree", "child": {"type": "mesh", "path": "./gem.obj", "material": mat()}}})
    children.append({"type": "transform", "matrix": H._trs(u(0, 6), (u(0.3, 0.8), u(0.3, 0.8), u(0.3, 0.8)), (u(-3, 3), u(-2, 1), u(-3, 1))),
                     "child": {"type": "mesh", "path": "./cube.obj", "material": mat()}})
    children.append({"type": "plane", "point": [0.0, -3.0, 0.0], "normal": [0.0, 1.0, 0.0], "tangent": [0.0, 0.0, -1.0],
                     "bitangent": [1.0, 0.0, 0.0], "material": mat()})
    children.append({"type": "plane", "point": [0.0, 0.0, -6.0], "normal": [0.0, 0.0, 1.0], "tangent": [0.0, 1.0, 0.0],
                     "bitangent": [1.0, 0.0, 0.0], "material": mat()})
    if rnd.random() < 0.5:
        children.append({"type": "plane", "point": [-6.0, 0.0, 0.0], "normal": [1.0, 0.0, 0.0], "tangent": [0.0, 1.0, 0.0],
                         "bitangent": [0.0, 0.0, -1.0], "material": mat()})
    return {"type": "node_collection", "children": children}


def random_camera(rnd):
    u = lambda lo, hi: float(rnd.uniform(lo, hi))  # noqa: E731
    yaw, pitch = u(-0.3, 0.3), u(-0.25, 0.25)
    ry = np.array([[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]])
    rx = np.array([[1, 0, 0], [0, math.cos(pitch), -math.sin(pitch)], [0, math.sin(pitch), math.cos(pitch)]])
    shape = ["point", "circle", "square"][int(rnd.integers(0, 3))]
    return P.Camera(resolution=[int(rnd.integers(5, 90)), int(rnd.integers(5, 70))], rotation=(ry @ rx).tolist(),
                    position=[u(-2, 2), u(-1, 3), u(10, 16)], fov=u(0.3, 1.1), near_plane=[0.0, u(0, 3)][int(rnd.integers(0, 2))],
                    focal_length=u(8, 18), bokeh_shape=shape, bokeh_size=[0.0, u(0, 0.6)][int(rnd.integers(0, 2))])


@pytest.fixture(scope="module")
def fuzz_root(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("fuzz_render"))
    shutil.rmtree(root)
    shutil.copytree(H.SCENE_DATA, root)
    with open(os.path.join(root, "gem.obj"), "w") as f:
        f.write(H.GEM_OBJ)
    open(os.path.join(root, "gem.tree"), "w").close()
    return root


@pytest.mark.parametrize("seed", range(int(os.environ.get("PATHTRACING_AMD_FUZZ_FIRST", "0")),
                                        int(os.environ.get("PATHTRACING_AMD_FUZZ_FIRST", "0")) + int(os.environ.get("PATHTRACING_AMD_FUZZ_SEEDS", "40"))))
def test_random_materials_cameras_options(oracle, fuzz_root, seed):
    rnd = np.random.default_rng(5000 + seed)
    for k in range(N_MATERIALS):
        with open(os.path.join(fuzz_root, "materials", f"r{k}.json"), "w") as f:
            json.dump(random_material(rnd), f)
    with open(os.path.join(fuzz_root, "fuzz.json"), "w") as f:
        json.dump(random_scene(rnd), f)
    scene = P.Scene.load(fuzz_root, "./fuzz.json")
    gpu = P.HipRenderer.init(scene, 0)
    cpu = oracle.scene(scene.desc_pointer)
    camera = random_camera(rnd)
    task = P.RenderTask(scene="./fuzz.json", config=P.Config(trace_depth=int(rnd.integers(1, 9)), iterations=int(rnd.integers(1, 5))),
                        camera=camera)
    w, h = camera.resolution
    opts = P.RenderOpts(seed=int(rnd.integers(0, 2 ** 40)), sample_offset=int(rnd.integers(0, 1000)) if rnd.random() < 0.5 else 0)
    if rnd.random() < 0.4:
        x0, y0 = int(rnd.integers(0, w)), int(rnd.integers(0, h))
        opts.tile = (x0, y0, int(rnd.integers(1, w + 8)), int(rnd.integers(1, h + 8)))
    if rnd.random() < 0.4:
        opts.paths_per_batch = int(rnd.integers(64, 5000))
    img = gpu.render(task, opts)
    ref, st = cpu.render(task, opts, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    assert gpu.last_stats.rays == st.rays and gpu.last_stats.paths == st.paths
    same = H.bits(img) == H.bits(ref)
    assert same.all(), f"seed {seed}: {np.count_nonzero(~same)} of {same.size} values differ, first at {np.argwhere(~same)[0]}"
    gpu.close()
    cpu.close()


@pytest.mark.parametrize("n_materials", [32, 33, 48])
def test_scenes_around_the_material_limit_of_the_fused_kernel(oracle, fuzz_root, n_materials):
    """A flat scene takes walk_bounce_kernel (materials from its LDS copy) while it has at most 32 materials and the
    walk and bounce kernels beyond: the same images as the oracle's on both sides of the limit, and through
    pt_trace_closest's wavefront route."""
    rnd = np.random.default_rng(77 + n_materials)
    u = lambda lo, hi: float(rnd.uniform(lo, hi))  # noqa: E731
    children = [{"type": "sphere", "center": [0.0, 7.0, 3.0], "radius": 3.0, "material": "./materials/mat0.json"}]
    for k in range(n_materials - 1):
        with open(os.path.join(fuzz_root, "materials", f"m{k}.json"), "w") as f:
            json.dump(random_material(rnd), f)
        children.append({"type": "sphere", "center": [u(-5, 5), u(-3, 3), u(-4, 2)], "radius": u(0.3, 0.9), "material": f"./materials/m{k}.json"})
    children.append({"type": "transform", "matrix": H._trs(0.7, (0.8,) * 3, (0.5, -1.0, 0.0)),
                     "child": {"type": "mesh", "path": "./gem.obj", "material": "./materials/m0.json"}})
    children.append({"type": "plane", "point": [0.0, -3.0, 0.0], "normal": [0.0, 1.0, 0.0], "tangent": [0.0, 0.0, -1.0],
                     "bitangent": [1.0, 0.0, 0.0], "material": "./materials/m1.json"})
    with open(os.path.join(fuzz_root, "many.json"), "w") as f:
        json.dump({"type": "node_collection", "children": children}, f)
    scene = P.Scene.load(fuzz_root, "./many.json")
    assert scene.desc.material_count == n_materials
    gpu = P.HipRenderer.init(scene, 0)
    cpu = oracle.scene(scene.desc_pointer)
    task = P.RenderTask(scene="./many.json", config=P.Config(trace_depth=6, iterations=3),
                        camera=P.Camera(resolution=[96, 64], rotation=np.eye(3).tolist(), position=[0.0, 1.0, 14.0], fov=0.9,
                                        near_plane=0.0, focal_length=12.0, bokeh_shape="circle", bokeh_size=0.2))
    opts = P.RenderOpts(seed=9)
    img = gpu.render(task, opts)
    ref, st = cpu.render(task, opts, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    assert gpu.last_stats.rays == st.rays
    assert (H.bits(img) == H.bits(ref)).all()
    rays = H.random_rays(5000, seed=n_materials)
    H.assert_all_routes(gpu, rays, cpu.trace(rays, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE), f"{n_materials} materials")
    gpu.close()
    cpu.close()
