"""The library's host loaders against oracle/loaders.py, an independent restatement of
Scene::load, the OBJ loader, the PNG decode and the material / task schemas (plain Python
written from the reference's source): every field of the loaded scene, for the shipped
scenes and for generated ones; and the oracle rendering a scene that only the restatement
has read gives the bits it gives for the library's reading of the same files."""
import json
import os

import numpy as np
import pytest

import pathtracing_amd as P
from oracle import loaders as L
from pathtracing_amd import _native as N
from tests import helpers as H
from tests import oracle_binding as ob
from tests.test_gpu_instances import write_run_scene

VEC_FIELDS = ("center", "point", "normal", "tangent", "bitangent")


def assert_same_scene(lib_scene: P.Scene, own: L.HostScene):
    d = lib_scene.desc
    assert d.node_count == len(own.nodes) and d.mesh_count == len(own.meshes)
    assert d.material_count == len(own.materials) and d.image_count == len(own.images) and d.tree_count == len(own.trees)
    for i, want in enumerate(own.nodes):
        got = d.nodes[i]
        assert (got.type, got.child_count) == (want["type"], want["child_count"]), i
        if want["type"] == N.NODE_TRANSFORM:
            assert [[got.matrix.m[r][c] for c in range(4)] for r in range(4)] == want["matrix"], i
        if want["type"] == N.NODE_MESH:
            assert (got.mesh_id, got.material_id) == (want["mesh_id"], want["material_id"]), i
        if want["type"] in (N.NODE_SPHERE, N.NODE_PLANE):
            assert got.material_id == want["material_id"], i
            assert got.radius == want["radius"]
            for name in VEC_FIELDS:
                v = getattr(got, name)
                assert [v.x, v.y, v.z] == want[name], (i, name)
        if want["type"] == N.NODE_KD_TREE:
            assert got.tree_id == want["tree_id"], i
    for i, (pos, nrm, valid, uvs) in enumerate(own.meshes):
        m = d.meshes[i]
        n = int(m.triangle_count)
        assert n == pos.shape[0]
        assert np.array_equal(np.ctypeslib.as_array(m.positions, shape=(n, 3, 3)).view(np.uint32), pos.view(np.uint32))
        got_valid = np.ctypeslib.as_array(m.normal_valid, shape=(n,)) if m.normal_valid else np.full(n, 1 if m.normals else 0, np.uint8)
        assert np.array_equal(got_valid != 0, valid != 0)
        if m.normals:
            got_n = np.ctypeslib.as_array(m.normals, shape=(n, 3, 3))
            assert np.array_equal(got_n[valid != 0], nrm[valid != 0])
        got_uv = np.ctypeslib.as_array(m.uvs, shape=(n, 3, 2)) if m.uvs else np.zeros((n, 3, 2), np.float32)
        assert np.array_equal(got_uv, uvs)
    for i, want in enumerate(own.materials):
        got = d.materials[i]
        assert got.type == want["type"], i
        if want["type"] == N.MATERIAL_BASE:
            assert got.color_is_texture == want["color_is_texture"]
            if want["color_is_texture"]:
                assert (got.image_id, got.uv_mode) == (want["image_id"], want["uv_mode"])
            else:
                assert [got.color.x, got.color.y, got.color.z] == want["color"]
            assert [got.emission.x, got.emission.y, got.emission.z] == want["emission"]
            for name in ("refraction", "reflective", "emissive", "refractive"):
                assert getattr(got, name) == want[name], (i, name)
        else:
            assert [got.albedo.x, got.albedo.y, got.albedo.z] == want["albedo"]
            assert (got.roughness, got.metallic) == (want["roughness"], want["metallic"])
    for i, want in enumerate(own.images):
        im = d.images[i]
        assert (im.height, im.width) == want.shape[:2]
        got = np.ctypeslib.as_array(im.rgba8, shape=(im.height, im.width, 4))
        assert np.array_equal(got[..., :3], want[..., :3]), i  # only rgb is ever read (image.rs:39-46)
    for i, want in enumerate(own.trees):
        assert d.trees[i].size == len(want)


@pytest.mark.parametrize("scene_path", ["./simple_scene.json", "./scene0.json", "./scene.json"])
def test_shipped_scenes_load_alike(scene_path):
    assert_same_scene(P.Scene.load(H.SCENE_DATA, scene_path), L.load_scene(H.SCENE_DATA, scene_path))


def test_generated_scenes_load_alike(tmp_path):
    d = str(tmp_path / "nested")
    path = H.write_nested_scene(d)
    assert_same_scene(P.Scene.load(d, path), L.load_scene(d, path))
    d = str(tmp_path / "run")
    path = write_run_scene(d)
    assert_same_scene(P.Scene.load(d, path), L.load_scene(d, path))


def test_obj_variants_load_alike(tmp_path):
    text = "\n".join([
        "# pentagon, negative indices, a point and a line that are dropped, CRLF, three models, numbers in every spelling",
        "o first", "v 0 0 0 1.0", "v 1e0 0 0", "v 1 1.000000001 0", "v .5 1.5 -0.0", "v 0 +1 0\r",
        "p 1", "l 1 2", "f -5 -4 -3 -2 -1",
        "g second", "vn 0 0 1", "vt 0.25 0.75 0.5", "f 1//1 2//1 3//1", "f 1/1/1 2/1/1 3/1/1",
        "o third", "vt 0.1 0.2", "f 1/2 3/1 4/2 5/1", ""])
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "materials"))
    with open(os.path.join(d, "materials", "m.json"), "w") as f:
        json.dump({"type": "pbr", "roughness": 0.3}, f)
    with open(os.path.join(d, "m.obj"), "w") as f:
        f.write(text)
    with open(os.path.join(d, "s.json"), "w") as f:
        json.dump({"type": "mesh", "path": "./m.obj", "material": "./materials/m.json"}, f)
    own = L.load_scene(d, "./s.json")
    assert own.meshes[0][0].shape[0] == 7 and own.meshes[0][2].tolist() == [0, 0, 0, 1, 1, 0, 0]
    assert_same_scene(P.Scene.load(d, "./s.json"), own)
    # a face that mixes corners with and without texture coordinates: the reference panics, both loaders refuse
    with open(os.path.join(d, "m.obj"), "w") as f:
        f.write(text.replace("f 1/1/1 2/1/1 3/1/1", "f 1/1/1 2//1 3/1/1"))
    with pytest.raises(ValueError):
        L.load_scene(d, "./s.json")
    with pytest.raises(N.PathTracingError):
        P.Scene.load(d, "./s.json")


def test_png_variants_decode_alike(tmp_path):
    from PIL import Image

    rng = np.random.default_rng(1)
    cases = {
        "rgb.png": Image.fromarray(rng.integers(0, 256, (37, 53, 3), dtype=np.uint8), "RGB"),
        "rgba.png": Image.fromarray(rng.integers(0, 256, (16, 9, 4), dtype=np.uint8), "RGBA"),
        "gray.png": Image.fromarray(rng.integers(0, 256, (20, 31), dtype=np.uint8), "L"),
        "pal.png": Image.fromarray(rng.integers(0, 7, (33, 29), dtype=np.uint8), "P"),
        "smooth.png": Image.fromarray((np.add.outer(np.arange(64), np.arange(80))[..., None] * [1, 2, 3] % 256).astype(np.uint8), "RGB"),
    }
    cases["pal.png"].putpalette([int(v) for v in rng.integers(0, 256, 21)])
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "materials"))
    for name, im in cases.items():
        im.save(os.path.join(d, name), optimize=(name == "pal.png"))
        with open(os.path.join(d, "materials", name + ".json"), "w") as f:
            json.dump({"type": "base", "color": {"type": "texture", "path": "./" + name, "uv_mode": "clamp"}}, f)
        ref = np.asarray(im.convert("RGBA"))
        assert np.array_equal(L.decode_png(open(os.path.join(d, name), "rb").read())[..., :3], ref[..., :3]), name
    children = [{"type": "sphere", "center": [0, 0, 0], "radius": 1, "material": f"./materials/{n}.json"} for n in cases]
    with open(os.path.join(d, "s.json"), "w") as f:
        json.dump({"type": "node_collection", "children": children}, f)
    assert_same_scene(P.Scene.load(d, "./s.json"), L.load_scene(d, "./s.json"))


def test_render_task_parses_alike():
    with open(os.path.join(H.SCENE_DATA, "render_task.json")) as f:
        text = f.read()
    t, own = P.RenderTask.from_json(text), L.parse_render_task(text)
    assert (t.scene, t.scene_md5, t.config.trace_depth, t.config.iterations) == (own["scene"], own["scene_md5"], own["trace_depth"], own["iterations"])
    cam = t.camera
    assert cam.resolution == own["resolution"] and cam.rotation == own["rotation"] and cam.position == own["position"]
    assert (cam.fov, cam.near_plane, cam.focal_length, cam.bokeh_shape, cam.bokeh_size) == (
        own["fov"], own["near_plane"], own["focal_length"], own["bokeh_shape"], own["bokeh_size"])


def test_oracle_renders_its_own_reading_of_the_files(oracle, tmp_path):
    """The oracle fed by the restated loaders alone (no product code has touched the scene)
    against the oracle fed by the library's loader: same rays, same image bits."""
    d = str(tmp_path / "nested")
    nested = H.write_nested_scene(d)
    for root, path in ((H.SCENE_DATA, "./simple_scene.json"), (H.SCENE_DATA, "./scene.json"), (d, nested)):
        own = L.SceneDesc(L.load_scene(root, path))
        a = oracle.scene(own.pointer)
        b = oracle.scene(P.Scene.load(root, path).desc_pointer)
        task = H.load_task()
        task.camera.resolution = [48, 36]
        task.config = P.Config(trace_depth=6, iterations=2)
        ia, sa = a.render(task, P.RenderOpts(seed=3), accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
        ib, sb = b.render(task, P.RenderOpts(seed=3), accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
        assert sa.rays == sb.rays and np.array_equal(H.bits(ia), H.bits(ib)) and ia.mean() > 0.01
        a.close()
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene_path", ["./simple_scene.json", "./scene0.json", "./scene.json"])
def test_gpu_against_the_oracle_on_its_own_loaders(oracle, scene_path):
    """End to end with nothing shared above the C ABI: the GPU renders what the library loaded,
    the oracle renders what oracle/loaders.py loaded; closest hits and image bit for bit."""
    own = L.SceneDesc(L.load_scene(H.SCENE_DATA, scene_path))
    cpu = oracle.scene(own.pointer)
    gpu = P.HipRenderer.init(P.Scene.load(H.SCENE_DATA, scene_path), 0)
    rays = H.random_rays(100_000, seed=31, origin_box=8.0, target_box=3.5)
    H.assert_all_routes(gpu, rays, cpu.trace(rays, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE), scene_path)
    task = H.load_task()
    task.scene = scene_path
    task.camera.resolution = [96, 64]
    task.config = P.Config(trace_depth=8, iterations=3)
    got = gpu.render(task, P.RenderOpts(seed=6))
    want, st = cpu.render(task, P.RenderOpts(seed=6), accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    assert gpu.last_stats.rays == st.rays and np.array_equal(H.bits(got), H.bits(want))
    gpu.close()
    cpu.close()
