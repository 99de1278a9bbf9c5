"""The C++ host mirror (include/pathtracing_amd.hpp) through its example program:
examples/render_task does what the reference worker does with one task."""
import json
import os
import subprocess

import numpy as np
import pytest

import pathtracing_amd as P
from tests import helpers as H

EXAMPLES = os.path.join(H.ROOT, "examples")
BINARY = os.path.join(EXAMPLES, "render_task")


def small_task(tmp_path, scene="./scene.json"):
    with open(os.path.join(H.SCENE_DATA, "render_task.json")) as f:
        task = json.load(f)
    task["scene"] = scene
    task["camera"]["resolution"] = [48, 32]
    task["config"] = {"trace_depth": 6, "iterations": 3}
    path = str(tmp_path / "task.json")
    with open(path, "w") as f:
        json.dump(task, f)
    return path


def test_example_builds_and_reports_errors(tmp_path):
    subprocess.check_call(["make", "-C", EXAMPLES], stdout=subprocess.DEVNULL)
    assert os.path.exists(BINARY)
    # a task whose scene file does not exist: the reference panics, this exits 1 with a message
    bad = small_task(tmp_path, scene="./no_such_scene.json")
    r = subprocess.run([BINARY, H.SCENE_DATA, bad, str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "no_such_scene" in r.stderr
    r = subprocess.run([BINARY], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_example_renders_the_same_bytes_as_the_python_worker(tmp_path):
    assert os.path.exists(BINARY), "examples/render_task is built by __graft_entry__.build()"
    task_path = small_task(tmp_path)
    out = str(tmp_path / "out.bin")
    r = subprocess.run([BINARY, H.SCENE_DATA, task_path, out, "11"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "48x32" in r.stdout
    with open(task_path) as f:
        task = P.RenderTask.from_json(f.read())
    image = P.Worker(H.SCENE_DATA).render(task, P.RenderOpts(seed=11))
    want = P.RenderedImage(image).to_bytes()
    with open(out, "rb") as f:
        got = f.read()
    assert got == want
    assert np.frombuffer(got[8:], dtype="<f4").reshape(32, 48, 3).mean() > 0
