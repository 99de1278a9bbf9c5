"""The benchmark configurations at their full sizes, checked through properties
that do not need a CPU render of the same size: splitting a frame by rows or by
sample range and summing gives the frame, a render repeats bit for bit, frame
means agree with a small oracle render of the same scene."""
import json
import os

import numpy as np
import pytest

import pathtracing_amd as P
from pathtracing_amd import distributed as D
from pathtracing_amd import synth
from tests import helpers as H
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu


def test_plumbing_config_256x256_4spp_against_the_literal_oracle(oracle):
    """Benchmark configuration 1 at its own size: simple_scene.json (the cube behind a
    transform, both textures, the emissive ceiling), 256x256, 4 samples, depth 8.  Bit for
    bit against the oracle in portable-math mode; against the literal restatement (libm,
    linear scan), whose paths branch differently after a last-place difference in sin/cos,
    statistically: RMSE(gpu, ref) <= 1.5 x RMSE(ref seed A, ref seed B), frame means within
    1 % (the sum over 65 k pixels x 4 samples)."""
    scene = P.Scene.load(H.SCENE_DATA, "./simple_scene.json")
    assert scene.desc.mesh_count == 1 and scene.desc.image_count == 2
    gpu = P.HipRenderer.init(scene, 0)
    cpu = oracle.scene(scene.desc_pointer)
    task = H.load_task()
    task.scene = "./simple_scene.json"
    task.camera.resolution = [256, 256]
    task.config = P.Config(trace_depth=8, iterations=4)
    got = gpu.render(task, P.RenderOpts(seed=1))
    same, st = cpu.render(task, P.RenderOpts(seed=1), accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    assert gpu.last_stats.rays == st.rays and np.array_equal(H.bits(got), H.bits(same))
    ref_a, _ = cpu.render(task, P.RenderOpts(seed=1), accel=ob.ACCEL_LINEAR, math=ob.MATH_LIBM)
    ref_b, _ = cpu.render(task, P.RenderOpts(seed=2), accel=ob.ACCEL_LINEAR, math=ob.MATH_LIBM)
    noise = H.rmse(ref_a, ref_b)
    assert H.rmse(got, ref_b) <= 1.5 * noise, (H.rmse(got, ref_b), noise)
    assert abs(got.mean() - ref_b.mean()) <= 0.01 * ref_b.mean() + 2.0 * noise / np.sqrt(got.size)
    gpu.close()
    cpu.close()


@pytest.mark.parametrize("scene_path", ["./simple_scene.json", "./scene0.json"])
def test_textured_scene_1080p_16spp(oracle, scene_path):
    """Benchmark configuration 2: 1920x1080, 16 samples, depth 8 on simple_scene.json --
    the scene with both textures (mat3 = test_texture_1.png on the cube, mat2 =
    test_texture_0.png on the walls) -- and, as a second case, on scene0.json (spheres and
    planes, one texture, the PBR material)."""
    scene = P.Scene.load(H.SCENE_DATA, scene_path)
    if scene_path == "./simple_scene.json":
        assert scene.desc.mesh_count == 1 and scene.desc.image_count == 2
    gpu = P.HipRenderer.init(scene, 0)
    task = H.load_task()
    task.scene = scene_path
    task.camera.resolution = [1920, 1080]
    task.config = P.Config(trace_depth=8, iterations=16)
    full = gpu.render(task, P.RenderOpts(seed=1))
    assert gpu.last_stats.paths == 1920 * 1080 * 16
    assert np.array_equal(H.bits(full), H.bits(gpu.render(task, P.RenderOpts(seed=1)))), "render does not repeat"
    # two batches in flight (a quarter of the frame's paths per batch) and one stream: the same bits
    small = P.RenderOpts(seed=1, paths_per_batch=1920 * 1080 * 4)
    assert np.array_equal(H.bits(full), H.bits(gpu.render(task, small)))
    small.single_stream = True
    assert np.array_equal(H.bits(full), H.bits(gpu.render(task, small)))
    # three row bands, as three ranks would render them, reassemble the frame bit for bit
    total = np.zeros_like(full)
    for rank in range(3):
        sub, opts = D.shard_opts(task, 3, rank, "rows", seed=1)
        total += gpu.render(sub, opts)
    assert np.array_equal(H.bits(total), H.bits(full))
    # two sample ranges: equal up to the order of the f32 sums
    halves = []
    for rank in range(2):
        sub, opts = D.shard_opts(task, 2, rank, "samples", seed=1)
        halves.append(gpu.render(sub, opts).astype(np.float64))
    assert np.allclose((halves[0] + halves[1]) / 2, full, rtol=2e-6, atol=1e-7)
    # frame mean against the oracle (literal mode: libm, linear scan) at 1/64 of the pixels;
    # both estimate the same integral, tolerance 2 %
    small_task = H.load_task()
    small_task.scene = scene_path
    small_task.camera.resolution = [240, 135]
    small_task.config = P.Config(trace_depth=8, iterations=16)
    cpu = oracle.scene(scene.desc_pointer)
    ref, _ = cpu.render(small_task, P.RenderOpts(seed=2), accel=ob.ACCEL_LINEAR, math=ob.MATH_LIBM)
    cpu.close()
    gpu.close()
    assert abs(full.mean() - ref.mean()) <= 0.02 * ref.mean()


@pytest.fixture(scope="module")
def dragon_dir(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dragon_cfg"))
    return out, synth.write_dragon_scene(out, H.SCENE_DATA)


def test_dragon_4k_256spp_in_eight_sample_ranges(dragon_dir):
    """The dragon at 3840x2160, 256 samples split into 8 ranges of 32 as 8 GPUs would
    take them (one GPU renders the ranges one after the other): their sum is the
    256-sample frame up to the order of the f32 additions."""
    out, task_path = dragon_dir
    task = H.load_task(task_path)
    task.camera.resolution = [3840, 2160]
    task.config = P.Config(trace_depth=8, iterations=256)
    gpu = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
    full = gpu.render(task, P.RenderOpts(seed=0))
    assert gpu.last_stats.paths == 3840 * 2160 * 256
    rays_full = gpu.last_stats.rays
    total = np.zeros(full.shape, dtype=np.float64)
    rays = 0
    for rank in range(8):
        sub, opts = D.shard_opts(task, 8, rank, "samples", seed=0)
        assert sub.config.iterations == 32 and opts.sample_offset == 32 * rank
        total += gpu.render(sub, opts)
        rays += gpu.last_stats.rays
    gpu.close()
    assert rays == rays_full  # the same paths were traced
    # 256 f32 additions per pixel in two different orders: a few 1e-5 relative at worst
    # (a path whose cosine-lobe sample hits the reference's sqrt(cos(pi/2)) < 0 case carries NaN
    # to its pixel, here as there: a handful of pixels per 2e9 paths, the same ones both ways)
    nan = np.isnan(full)
    assert np.array_equal(nan, np.isnan(total)) and nan.mean() < 1e-4
    err = np.abs(total / 8 - full)[~nan] / np.maximum(np.abs(full[~nan]), 1e-3)
    # (multipliers of the PBR lobes may be large and of either sign, so single pixels cancel)
    assert err.max() < 2e-3 and np.quantile(err, 0.999) < 2e-5, (err.max(), np.quantile(err, 0.999))


def test_instanced_dragons_through_the_scene_graph(oracle, dragon_dir):
    """Nine transformed instances of the 871k-triangle mesh (7.8 M triangles on
    screen, one BVH in memory): closest hits against the oracle, which walks the same
    scene graph.  More mesh nodes than the search kernel takes ahead of the walk, so
    the walk kernel's own search is exercised on a large BVH as well."""
    out, _ = dragon_dir
    with open(os.path.join(out, "instanced.json"), "w") as f:
        json.dump(synth.instanced_dragon_scene_json(9, seed=3), f)
    scene = P.Scene.load(out, "./instanced.json")
    gpu = P.HipRenderer.init(scene, 0)
    cpu = oracle.scene(scene.desc_pointer)
    rng = np.random.default_rng(17)
    n = 200_000
    src = np.stack([rng.uniform(-38, 38, n), rng.uniform(-30, 38, n), rng.uniform(-38, 38, n)], axis=1)
    dst = np.stack([rng.uniform(-35, 35, n), rng.uniform(-14, 10, n), rng.uniform(-35, 35, n)], axis=1)
    d = dst - src
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 7] = src, d, H.F32_MAX
    want = cpu.trace(rays, accel=ob.ACCEL_BVH, math=ob.MATH_PORTABLE)
    H.assert_all_routes(gpu, rays, want, "instanced dragons")
    on_mesh = (want["hit"] == 1) & (want["tri_index"] != 0xFFFFFFFF)
    assert on_mesh.mean() > 0.2 and np.unique(want["node_index"][on_mesh]).size == 9
    # a small render through the wavefront pipeline, bit for bit
    task = H.load_task()
    task.camera.resolution = [96, 64]
    task.camera.position = [0.0, 10.0, 110.0]
    task.camera.focal_length = 110.0
    task.config = P.Config(trace_depth=6, iterations=2)
    got = gpu.render(task, P.RenderOpts(seed=4))
    ref, st = cpu.render(task, P.RenderOpts(seed=4), accel=ob.ACCEL_BVH, math=ob.MATH_PORTABLE)
    assert gpu.last_stats.rays == st.rays
    assert np.array_equal(H.bits(got), H.bits(ref))
    gpu.close()
    cpu.close()


def test_flattened_instances_seven_million_triangles(oracle, dragon_dir):
    """The instanced workload the other way round: eight copies of the 871k-triangle
    mesh flattened into one mesh of 7.0 M world-space triangles behind the dragon
    scene's graph (one BVH of 2.9 M leaves).  Closest hits against the oracle's own
    BVH over the same triangles, and the tree's structural check."""
    import ctypes as C

    from pathtracing_amd import _native as N
    from pathtracing_amd.scene import build_tree, validate_tree

    out, task_path = dragon_dir
    task = H.load_task(task_path)
    scene = P.Scene.load(out, task.scene)
    base = scene.desc
    assert base.mesh_count == 1
    n1 = int(base.meshes[0].triangle_count)
    one = np.ctypeslib.as_array(base.meshes[0].positions, shape=(n1, 9))
    copies = []
    for i in range(8):
        shift = np.array([26.0 * (i % 4 - 1.5), 0.0, -40.0 * (i // 4)], dtype=np.float32)
        copies.append(one + np.tile(shift, 3))
    positions = np.ascontiguousarray(np.concatenate(copies), dtype=np.float32)
    mesh = N.PtMeshDesc()
    mesh.triangle_count = positions.shape[0]
    mesh.positions = positions.ctypes.data_as(C.POINTER(C.c_float))
    big = N.PtSceneDesc()
    for name, _ in N.PtSceneDesc._fields_:
        setattr(big, name, getattr(base, name))
    big.meshes = C.pointer(mesh)
    big.mesh_count = 1
    # the tree the library builds passes its own structural and conservativeness check
    blob = build_tree(mesh)
    assert validate_tree(mesh, blob)
    gpu = P.HipRenderer.from_desc(C.pointer(big), 0)
    cpu = oracle.scene(C.pointer(big))
    rng = np.random.default_rng(29)
    n = 300_000
    src = np.stack([rng.uniform(-60, 60, n), rng.uniform(-14, 14, n), rng.uniform(-60, 70, n)], axis=1)
    dst = np.stack([rng.uniform(-50, 50, n), rng.uniform(-12, 8, n), rng.uniform(-50, 10, n)], axis=1)
    d = dst - src
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 7] = src, d, H.F32_MAX
    want = cpu.trace(rays, accel=ob.ACCEL_BVH, math=ob.MATH_PORTABLE)
    H.assert_all_routes(gpu, rays, want, "flattened instances")
    on_mesh = (want["hit"] == 1) & (want["tri_index"] != 0xFFFFFFFF)
    assert on_mesh.mean() > 0.2
    assert np.unique(want["tri_index"][on_mesh] // n1).size == 8  # every copy is hit
    gpu.close()
    cpu.close()
