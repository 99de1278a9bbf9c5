"""The worker's front end without sockets: postcard bytes of the discovery messages, MD5,
the client's scene digest, and (on the GPU) one round of connection_loop -- RenderTask
text in, RenderedImage bytes out -- over directory buckets named by scene_md5."""
import ctypes as C
import hashlib
import json
import os
import shutil

import numpy as np
import pytest

import pathtracing_amd as P
from pathtracing_amd import _native as N
from pathtracing_amd import front_end as F
from oracle import loaders as L
from tests import helpers as H


def test_discovery_messages_in_postcard_bytes():
    # worker/src/lib.rs:81-91 under postcard 1.1: a struct without fields is no bytes; u16 is an LEB128 varint
    assert F.discovery_request_bytes() == b""
    assert F.discovery_request_ok(b"") and F.discovery_request_ok(b"\x00anything")
    known = {0: "00", 1: "01", 127: "7f", 128: "8001", 300: "ac02", 16383: "ff7f", 16384: "808001", 30000: "b0ea01",
             40000: "c0b802", 65535: "ffff03"}
    for port, hexed in known.items():
        assert F.discovery_response_bytes(port).hex() == hexed, port
        assert F.discovery_response_port(bytes.fromhex(hexed)) == port
        assert F.discovery_response_port(bytes.fromhex(hexed) + b"\xff\xff") == port  # trailing bytes are ignored
    assert F.discovery_response_bytes() == bytes.fromhex("b0ea01")  # WEBSOCKET_PORT, main.rs:17
    for port in range(0, 65536, 97):
        assert F.discovery_response_port(F.discovery_response_bytes(port)) == port
    for bad in ("", "80", "8080", "ffff04", "ffff7f", "80808001"):
        with pytest.raises(N.PathTracingError) as e:
            F.discovery_response_port(bytes.fromhex(bad))
        assert e.value.status == 4  # PT_ERR_PARSE


def _py_scene_md5(root, scene_path):
    """client/src/scene.rs:19-85 with hashlib over oracle/loaders.py's reading of the references."""
    own = L.load_scene(root, scene_path)
    paths = sorted({scene_path, *own.mesh_paths, *own.material_paths, *own.image_paths})
    acc = ""
    for p in paths:
        with open(os.path.join(root, p), "rb") as f:
            acc = hashlib.md5((acc + hashlib.md5(f.read()).hexdigest()).encode()).hexdigest()
    return acc


@pytest.mark.parametrize("scene_path", ["./simple_scene.json", "./scene0.json", "./scene.json"])
def test_scene_digest_follows_the_client(scene_path):
    got = F.scene_md5(H.SCENE_DATA, scene_path)
    assert len(got) == 32 and got == got.lower() and got == _py_scene_md5(H.SCENE_DATA, scene_path)


def test_scene_digest_changes_with_any_file(tmp_path):
    d = str(tmp_path / "s")
    shutil.copytree(H.SCENE_DATA, d)
    before = F.scene_md5(d, "./simple_scene.json")
    with open(os.path.join(d, "materials", "mat3.json"), "a") as f:
        f.write("\n")
    after = F.scene_md5(d, "./simple_scene.json")
    assert after != before and after == _py_scene_md5(d, "./simple_scene.json")
    with pytest.raises(N.PathTracingError) as e:
        F.scene_md5(d, "./missing.json")
    assert e.value.status == 3


@pytest.mark.gpu
def test_connection_loop_round_over_md5_buckets(tmp_path):
    store = str(tmp_path / "store")
    os.makedirs(store)
    md5 = F.scene_md5(H.SCENE_DATA, "./scene.json")
    shutil.copytree(H.SCENE_DATA, os.path.join(store, md5))  # the bucket of this scene (file_store.rs:17-23)
    with open(os.path.join(H.SCENE_DATA, "render_task.json")) as f:
        wire = json.load(f)
    wire["scene"] = "./scene.json"
    wire["scene_md5"] = md5
    wire["camera"]["resolution"] = [80, 56]
    wire["config"] = {"trace_depth": 6, "iterations": 3}
    text = json.dumps(wire)
    worker = F.WorkerFrontEnd(store, 0)
    frame = worker.handle_message(text, P.RenderOpts(seed=4))
    assert worker.cached_scene_count == 1
    # the same task through the renderer directly
    task = P.RenderTask.from_json(text)
    want = P.HipRenderer.init(P.Scene.load(H.SCENE_DATA, "./scene.json"), 0).render(task, P.RenderOpts(seed=4))
    assert frame == P.RenderedImage(want).to_bytes()
    assert frame[:8] == (80).to_bytes(4, "little") + (56).to_bytes(4, "little") and want.mean() > 0.01
    # the first load wrote the bucket's missing .tree (scene.json: transform -> kd_tree -> mesh, cube.tree empty upstream)
    assert os.path.getsize(os.path.join(store, md5, "cube.tree")) > 64
    # a second task for the same scene_md5 is served from the cache: the bucket may be gone
    shutil.rmtree(os.path.join(store, md5))
    wire["config"]["iterations"] = 1
    again = worker.handle_message(json.dumps(wire), P.RenderOpts(seed=4))
    assert worker.cached_scene_count == 1 and len(again) == len(frame)
    # a frame that does not decode: the reference answers with this error and drops the connection
    with pytest.raises(N.PathTracingError) as e:
        worker.handle_message('{"scene": "./scene.json"}')
    assert e.value.status == 4 and "Failed to decode render task" in e.value.message
    # an unknown bucket: the files cannot be fetched
    wire["scene_md5"] = "0" * 32
    with pytest.raises(N.PathTracingError) as e:
        worker.handle_message(json.dumps(wire))
    assert e.value.status == 3
    # the file form (no scene_md5) is not a RenderTask: serde refuses it upstream (render_task.rs:23-29, main.rs:79)
    del wire["scene_md5"]
    with pytest.raises(N.PathTracingError) as e:
        worker.handle_message(json.dumps(wire))
    assert e.value.status == 4 and "Failed to decode render task" in e.value.message and "scene_md5" in e.value.message
    worker.close()


@pytest.mark.gpu
def test_names_in_a_frame_stay_inside_the_bucket(tmp_path):
    """The scene name of a task frame, and the names inside the files it leads to, come from a peer:
    none of them may read or write outside store/<scene_md5>/."""
    store = str(tmp_path / "store")
    md5 = "ab" * 16
    bucket = os.path.join(store, md5)
    shutil.copytree(H.SCENE_DATA, bucket)
    outside = tmp_path / "outside"
    shutil.copytree(H.SCENE_DATA, outside)
    with open(os.path.join(H.SCENE_DATA, "render_task.json")) as f:
        wire = json.load(f)
    wire["scene_md5"] = md5
    wire["camera"]["resolution"] = [32, 24]
    wire["config"] = {"trace_depth": 2, "iterations": 1}
    worker = F.WorkerFrontEnd(store, 0)
    for name in ("../../outside/scene.json", "/etc/passwd", "./a/../../x.json", "..", str(outside / "scene.json")):
        wire["scene"] = name
        with pytest.raises(N.PathTracingError) as e:
            worker.handle_message(json.dumps(wire))
        assert e.value.status == 1 and "leaves the scene's bucket" in e.value.message, name
    # a scene file inside the bucket whose kd_tree and mesh names point outside it
    with open(os.path.join(bucket, "scene.json")) as f:
        text = f.read()
    assert "./cube.tree" in text and "./cube.obj" in text
    with open(os.path.join(bucket, "tree_outside.json"), "w") as f:
        f.write(text.replace("./cube.tree", "../../outside/written.tree"))
    with open(os.path.join(bucket, "mesh_outside.json"), "w") as f:
        f.write(text.replace("./cube.obj", "../../outside/cube.obj"))
    for name in ("./tree_outside.json", "./mesh_outside.json"):
        wire["scene"] = name
        with pytest.raises(N.PathTracingError) as e:
            worker.handle_message(json.dumps(wire))
        assert e.value.status == 1, name
    assert not os.path.exists(outside / "written.tree") and worker.cached_scene_count == 0
    # names with dots that stay inside are fine
    wire["scene"] = "./scene.json"
    assert len(worker.handle_message(json.dumps(wire))) == 8 + 32 * 24 * 12
    worker.close()
