"""pt_render_multi / pt_shard: several GPUs driven from one process.  On CPU: the library's
split equals pathtracing_amd.distributed's (the one-process-per-GPU path).  On the GPU box
(one GPU): one device equals pt_render bit for bit; two scene handles on the same device
exercise the threads, the per-device streams and the peer-copy reduce (RCCL needs distinct
devices, which a one-GPU box does not have)."""
import numpy as np
import pytest

import pathtracing_amd as P
from pathtracing_amd import distributed as D
from pathtracing_amd import multi as M
from tests import helpers as H


def test_library_split_equals_the_python_split():
    task = H.load_task()
    for res in ([64, 45], [1920, 1080], [40, 8], [33, 3]):
        task.camera.resolution = list(res)
        for iters in (0, 1, 5, 64, 257):
            task.config.iterations = iters
            for world in (1, 2, 3, 8):
                for rank in range(world):
                    base = P.RenderOpts(seed=7, sample_offset=11)
                    sub, o = D.shard_opts(task, world, rank, "samples", seed=7, base=base)
                    it, off, _ = M.shard(task, base, "samples", world, rank)
                    if o is None:
                        assert it == 0
                    else:
                        assert (it, off) == (sub.config.iterations, o.sample_offset)
                    sub, o = D.shard_opts(task, world, rank, "rows", seed=7, base=base)
                    it, off, tile = M.shard(task, base, "rows", world, rank)
                    if o is None:
                        assert it == 0 or tile[3] == 0
                    else:
                        assert tuple(tile) == tuple(o.tile) and off == 11


@pytest.mark.gpu
def test_one_device_is_pt_render_and_two_handles_sum():
    scene = P.Scene.load(H.SCENE_DATA, "./simple_scene.json")
    a = P.HipRenderer.init(scene, 0)
    b = P.HipRenderer.init(scene, 0)
    task = H.load_task()
    task.scene = "./simple_scene.json"
    task.camera.resolution = [322, 187]
    task.config = P.Config(trace_depth=6, iterations=9)
    want = a.render(task, P.RenderOpts(seed=5))
    rays = a.last_stats.rays
    for mode in ("samples", "rows"):
        got, st = P.render_multi([a], task, P.RenderOpts(seed=5), mode)
        assert np.array_equal(H.bits(got), H.bits(want)) and st.rays == rays
    # two handles: rows reassemble bit for bit, sample ranges up to the order of the f32 sums
    got, st = P.render_multi([a, b], task, P.RenderOpts(seed=5), "rows")
    assert np.array_equal(H.bits(got), H.bits(want)) and st.rays == rays
    got, st = P.render_multi([a, b], task, P.RenderOpts(seed=5), "samples")
    assert st.rays == rays and st.paths == 322 * 187 * 9
    assert np.allclose(got, want, rtol=2e-6, atol=1e-7)
    # three handles, more handles than samples: one lane has nothing to render
    task.config.iterations = 2
    want2 = a.render(task, P.RenderOpts(seed=5))
    got, _ = P.render_multi([a, b, a], task, P.RenderOpts(seed=5), "samples")
    assert np.allclose(got, want2, rtol=2e-6, atol=1e-7)
    a.close()
    b.close()


@pytest.mark.gpu
def test_buffers_kept_between_calls_and_the_image_left_on_the_device():
    """The streams and buffers of a device list outlive the call: a larger frame after a smaller one,
    a smaller one after that, the image resolved straight into device memory, the buffers released
    and allocated again -- each against pt_render.  Counters asked for in `opts` are summed over the
    devices; a frame without samples and a tile together with row bands are refused."""
    import torch

    from pathtracing_amd import _native as N

    scene = P.Scene.load(H.SCENE_DATA, "./simple_scene.json")
    a = P.HipRenderer.init(scene, 0)
    b = P.HipRenderer.init(scene, 0)
    task = H.load_task()
    task.scene = "./simple_scene.json"
    for res, iters in (([96, 64], 4), ([322, 187], 5), ([40, 24], 3)):
        task.camera.resolution = list(res)
        task.config = P.Config(trace_depth=5, iterations=iters)
        want = a.render(task, P.RenderOpts(seed=9))
        got, st = P.render_multi([a, b], task, P.RenderOpts(seed=9), "rows")
        assert np.array_equal(H.bits(got), H.bits(want)) and st.rays == a.last_stats.rays
        assert 0.0 <= st.reduce_ms <= st.render_ms
        dev = torch.full((res[1] * res[0] * 3,), float("nan"), dtype=torch.float32, device="cuda:0")
        none, st = P.render_multi([a, b], task, P.RenderOpts(seed=9), "rows", out_ptr=dev.data_ptr(), width_height=tuple(res))
        torch.cuda.synchronize()
        assert none is None and np.array_equal(H.bits(dev.cpu().numpy().reshape(res[1], res[0], 3)), H.bits(want))
        N.check(N.lib.pt_device_release_workspace(0))
    # counters: what one device counts for the frame is what two count for its halves
    o = P.RenderOpts(seed=9)
    o.count_visits = True
    a.render(task, o)
    one = a.last_stats
    _, two = P.render_multi([a, b], task, o, "rows")
    assert one.node_visits > 0 and (two.node_visits, two.triangle_tests, two.searches) == (one.node_visits, one.triangle_tests, one.searches)
    # refusals
    task.config.iterations = 0
    with pytest.raises(N.PathTracingError) as e:
        P.render_multi([a, b], task, P.RenderOpts(seed=9), "samples")
    assert e.value.status == 1
    task.config.iterations = 2
    tiled = P.RenderOpts(seed=9)
    tiled.tile = (0, 0, 8, 8)
    with pytest.raises(N.PathTracingError) as e:
        P.render_multi([a, b], task, tiled, "rows")
    assert e.value.status == 1
    got, _ = P.render_multi([a, b], task, tiled, "samples")  # a tile with sample ranges is fine
    assert np.allclose(got, a.render(task, tiled), rtol=2e-6, atol=1e-7)
    a.close()
    b.close()
