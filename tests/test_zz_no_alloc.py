"""pt_trace_closest_device_static -- the closest-hit query that allocates nothing (its
scratch is a static device array) -- against pt_trace_closest, bit for bit, on two calls
in a row, the second with a smaller input.  Collected last: a failure here must not
hide the tests before it."""
import ctypes as C

import numpy as np
import pytest

import pathtracing_amd as P
from pathtracing_amd import _native as N
from pathtracing_amd import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _static_trace(gpu, rays: np.ndarray, stream=0, brute=False) -> np.ndarray:
    """Device buffers come from torch (the caller's business); the entry point itself gets
    plain pointers and sizes."""
    import torch

    d_rays = torch.from_numpy(np.ascontiguousarray(rays, dtype=np.float32)).to("cuda:0")
    d_hits = torch.zeros(rays.shape[0] * P.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    N.check(N.lib.pt_trace_closest_device_static(gpu._handle, d_rays.data_ptr(), rays.shape[0],
                                                 N.PT_FLAG_BRUTE_FORCE if brute else 0, d_hits.data_ptr(), stream))
    torch.cuda.synchronize()
    return d_hits.cpu().numpy().view(P.HIT_DTYPE)


def _same(got, want, what):
    H.assert_hits_identical(got, want, what)
    hit = want["hit"] == 1
    assert np.array_equal(got[hit].tobytes(), want[hit].tobytes()), f"{what}: hit records differ in some byte"


def test_static_trace_equals_pt_trace_closest(tmp_path):
    # nested collections and transforms (frames saved in the scratch columns), kd_tree nodes, glass
    d = str(tmp_path / "nested")
    gpu = P.HipRenderer.init(P.Scene.load(d, H.write_nested_scene(d)), 0)
    first = H.random_rays(200_000, seed=21, origin_box=9.0, target_box=4.0)
    second = H.random_rays(37_001, seed=22, origin_box=7.0, target_box=3.0)  # smaller, and not a multiple of 64
    got1 = _static_trace(gpu, first)
    got2 = _static_trace(gpu, second)
    _same(got1, gpu.trace_closest(first), "first call")
    _same(got2, gpu.trace_closest(second), "second, smaller call")
    assert got1["hit"].mean() > 0.3 and got2["hit"].mean() > 0.3
    # the linear-scan flag, a non-default stream, an empty input
    import torch

    st = torch.cuda.Stream()
    _same(_static_trace(gpu, second[:5000], stream=st.cuda_stream, brute=True), gpu.trace_closest(second[:5000], brute_force=True),
          "linear scan on a stream")
    N.check(N.lib.pt_trace_closest_device_static(gpu._handle, None, 0, 0, None, None))
    # sizes beyond the maximum and unknown flags are refused before anything is touched
    assert N.lib.pt_trace_closest_device_static(gpu._handle, C.c_void_p(16), N.PT_STATIC_TRACE_MAX_RAYS + 1, 0, C.c_void_p(16), None) == 1
    assert "PT_STATIC_TRACE_MAX_RAYS" in N.last_error()
    assert N.lib.pt_trace_closest_device_static(gpu._handle, C.c_void_p(16), 1, N.PT_FLAG_WAVEFRONT, C.c_void_p(16), None) == 1
    gpu.close()


def test_static_trace_deep_searches(tmp_path):
    """The 871k-triangle mesh: searches whose pending-node stack outgrows its 16 entries in
    LDS continue in the scratch columns."""
    out = str(tmp_path)
    task = H.load_task(synth.write_dragon_scene(out, H.SCENE_DATA))
    gpu = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
    rng = np.random.default_rng(5)
    n = 150_000
    px = np.stack([rng.integers(0, 1920, n), rng.integers(0, 1080, n)], axis=1).astype(np.uint32)
    rays = P.generate_rays(task.camera, 0, px, rng.integers(0, 64, n).astype(np.uint64))
    want = gpu.trace_closest(rays)
    sec = H.secondary_rays(want, rays, seed=6)[:40_000]
    _same(_static_trace(gpu, rays), want, "dragon primary")
    _same(_static_trace(gpu, sec), gpu.trace_closest(sec), "dragon secondary, smaller call")
    gpu.close()
