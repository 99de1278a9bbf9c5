"""The flattened-instances configuration at its full size: 115 transformed copies
of the 871k-triangle mesh as ONE mesh of 1.0e8 world-space triangles (a 5.8 GB
search structure), 3840x2160.

The copy count comes from PATHTRACING_AMD_FULL_SIZE_COPIES (default 115, the full
size: about two minutes on the GPU box; a smaller count shortens a local run).
Checked through what does not need a CPU render of that size: closest hits
against the oracle's own BVH over the same triangles, against the linear scan on
the GPU and (a few rays) the oracle's linear scan, and a frame split into eight
row bands reassembling bit for bit.
"""
import ctypes as C
import json
import os
import time

import numpy as np
import pytest

import pathtracing_amd as P
from pathtracing_amd import _native as N
from pathtracing_amd import distributed as D
from pathtracing_amd import synth
from tests import helpers as H
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu

COPIES = int(os.environ.get("PATHTRACING_AMD_FULL_SIZE_COPIES", "115"))


def _progress():
    """Progress lines for a run that is silent for minutes: stdout (visible with
    -s) and, when the directory exists, a log file the launcher watches."""
    t0 = time.time()
    log = os.path.join(H.ROOT, "gpurun_out", "full_size.log") if os.path.isdir(os.path.join(H.ROOT, "gpurun_out")) else None

    def say(msg):
        line = f"[{time.time() - t0:7.1f}s] {msg}"
        print(line, flush=True)
        if log:
            with open(log, "a") as f:
                f.write(line + "\n")

    return say


@pytest.mark.timeout(1500)
def test_flattened_instances_at_full_size(oracle, tmp_path):
    say = _progress()
    out = str(tmp_path)
    task = H.load_task(synth.write_dragon_scene(out, H.SCENE_DATA))
    with open(os.path.join(out, "flattened.json"), "w") as f:
        json.dump(synth.flattened_scene_json(), f)
    task.scene = "./flattened.json"
    task.camera.resolution = [3840, 2160]
    task.camera.position = [0.0, 0.0, 330.0]
    task.camera.focal_length = 330.0
    task.config = P.Config(trace_depth=8, iterations=1)
    scene = P.Scene.load(out, task.scene)
    base = scene.desc
    n1 = int(base.meshes[0].triangle_count)
    one = np.ctypeslib.as_array(base.meshes[0].positions, shape=(n1, 9))
    positions = synth.flattened_copies(one, COPIES, seed=5)
    say(f"{positions.shape[0]} triangles")
    mesh = N.PtMeshDesc()
    mesh.triangle_count = positions.shape[0]
    mesh.positions = positions.ctypes.data_as(C.POINTER(C.c_float))
    big = N.PtSceneDesc()
    for name, _ in N.PtSceneDesc._fields_:
        setattr(big, name, getattr(base, name))
    big.meshes = C.pointer(mesh)
    big.mesh_count = 1
    gpu = P.HipRenderer.from_desc(C.pointer(big), 0)
    say("search structure built and uploaded")
    cpu = oracle.scene(C.pointer(big))

    rng = np.random.default_rng(31)
    n, ext = 200_000, 80.0
    src = rng.uniform(-ext, ext, (n, 3))
    dst = rng.uniform(-0.7 * ext, 0.7 * ext, (n, 3))
    d = dst - src
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 7] = src, d, H.F32_MAX
    got = gpu.trace_closest(rays)
    on_mesh = (got["hit"] == 1) & (got["tri_index"] != 0xFFFFFFFF)
    assert on_mesh.mean() > 0.2
    assert np.unique(got["tri_index"][on_mesh] // n1).size == COPIES  # every copy is hit
    want = cpu.trace(rays, accel=ob.ACCEL_BVH, math=ob.MATH_PORTABLE)
    say("oracle BVH search done")
    H.assert_hits_identical(got, want, "flattened instances, oracle BVH")
    for route in ("wavefront", "wavefront_split"):
        H.assert_hits_identical(gpu.trace_closest(rays, route=route), want, f"flattened instances, oracle BVH [{route}]")
    # the linear scan: every lane walks all triangles, so the ray count does not matter much
    scan = gpu.trace_closest(rays[:1024], brute_force=True)
    say("GPU linear scan done")
    H.assert_hits_identical(got[:1024], scan, "flattened instances, GPU linear scan")
    few = cpu.trace(rays[:32], accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    say("oracle linear scan done")
    H.assert_hits_identical(got[:32], few, "flattened instances, oracle linear scan")
    cpu.close()

    full = gpu.render(task, P.RenderOpts(seed=1))
    assert gpu.last_stats.paths == 3840 * 2160
    total = np.zeros_like(full)
    for rank in range(8):
        sub, opts = D.shard_opts(task, 8, rank, "rows", seed=1)
        total += gpu.render(sub, opts)
    assert np.array_equal(H.bits(total), H.bits(full))
    say("row bands reassemble the frame")
    gpu.close()
