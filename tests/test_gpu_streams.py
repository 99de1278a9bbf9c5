"""One scene, several streams; several scenes, one device.  The queues of the wavefront
pipeline belong to the device and are handed from call to call through events, so
renders and traces queued from different streams (or on different scenes) must come
out as if they had run one after the other."""
import ctypes as C

import numpy as np
import pytest

import pathtracing_amd as P
from pathtracing_amd import _native as N
from tests import helpers as H
from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu


def test_two_streams_one_scene_equal_sequential():
    import torch

    scene = P.Scene.load(H.SCENE_DATA, "./simple_scene.json")
    gpu = P.HipRenderer.init(scene, 0)
    task = H.load_task()
    task.scene = "./simple_scene.json"
    task.camera.resolution = [640, 360]
    task.config = P.Config(trace_depth=6, iterations=8)
    n = 640 * 360 * 3
    # sequential, on the default stream
    want = []
    for seed in (1, 2, 3, 4):
        acc = torch.zeros(n, dtype=torch.float32, device="cuda:0")
        gpu.render_accumulate_device(task, P.RenderOpts(seed=seed), acc.data_ptr(), 0)
        torch.cuda.synchronize()
        want.append(acc.cpu().numpy())
    # the same four renders queued back to back on two non-blocking streams, no host wait in between;
    # small batches so that each render also keeps two batches in flight
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    accs = [torch.zeros(n, dtype=torch.float32, device="cuda:0") for _ in range(4)]
    torch.cuda.synchronize()
    for i, seed in enumerate((1, 2, 3, 4)):
        st = (s1, s2)[i % 2]
        gpu.render_accumulate_device(task, P.RenderOpts(seed=seed, paths_per_batch=640 * 360 * 2), accs[i].data_ptr(), st.cuda_stream)
    rays = H.random_rays(50_000, seed=9, origin_box=8.0, target_box=3.0)
    hits_during = gpu.trace_closest(rays)  # default stream, while the renders are queued
    hits_wave = gpu.trace_closest(rays, route="wavefront")
    torch.cuda.synchronize()
    for i in range(4):
        assert np.array_equal(accs[i].cpu().numpy().view(np.uint32), want[i].view(np.uint32)), f"render {i}"
    H.assert_hits_identical(hits_during, gpu.trace_closest(rays), "trace while renders are queued")
    H.assert_hits_identical(hits_wave, hits_during, "wavefront trace while renders are queued")
    gpu.close()


def test_scenes_of_one_device_share_the_workspace(oracle):
    """Two scenes with different demands (one mesh node / six mesh nodes) rendering in
    turns through the same workspace, and the workspace released in between."""
    a = P.Scene.load(H.SCENE_DATA, "./simple_scene.json")
    ga = P.HipRenderer.init(a, 0)
    ta = H.load_task()
    ta.scene = "./simple_scene.json"
    ta.camera.resolution = [96, 64]
    ta.config = P.Config(trace_depth=6, iterations=2)
    ca = oracle.scene(a.desc_pointer)
    want_a, _ = ca.render(ta, P.RenderOpts(seed=3), accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    b = P.Scene.load(H.SCENE_DATA, "./scene0.json")
    gb = P.HipRenderer.init(b, 0)
    tb = H.load_task()
    tb.scene = "./scene0.json"
    tb.camera.resolution = [128, 80]
    tb.config = P.Config(trace_depth=6, iterations=3)
    cb = oracle.scene(b.desc_pointer)
    want_b, _ = cb.render(tb, P.RenderOpts(seed=3), accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE)
    for round_ in range(3):
        assert np.array_equal(H.bits(ga.render(ta, P.RenderOpts(seed=3))), H.bits(want_a))
        assert np.array_equal(H.bits(gb.render(tb, P.RenderOpts(seed=3))), H.bits(want_b))
        if round_ == 1:
            N.check(N.lib.pt_device_release_workspace(0))
    for g in (ga, gb):
        g.close()
    for c in (ca, cb):
        c.close()
