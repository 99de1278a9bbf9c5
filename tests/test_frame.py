"""The client's frame (client/src/frame.rs): sum of renders, mean image, RGBA8
preview.  Oracle known answers on the CPU; GPU against the oracle bit for bit."""
import numpy as np
import pytest

import pathtracing_amd as P
from tests import helpers as H


def special_renders(h, w, count, seed):
    """Renders with the values a path tracer produces plus the cast's corner cases."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        img = rng.gamma(0.6, 2.0, size=(h, w, 3)).astype(np.float32)
        flat = img.reshape(-1)
        picks = rng.choice(flat.size, size=min(flat.size, 64), replace=False)
        specials = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, -0.5, -1.0, -2.0, 1e30, 3.4e38, 1e-30, 254.9999 / 0.0001,
                             1.0, 0.003937, 0.0039215689], dtype=np.float32)
        flat[picks] = specials[np.arange(picks.size) % specials.size]
        out.append(img)
    return out


def test_frame_known_answers(oracle):
    # frame.rs:91-93 by hand: c/(1+c)*255 truncated; mean of two renders first (frame.rs:73-83)
    a = np.array([[[0.0, 1.0, 3.0], [np.nan, -0.25, 1e30]]], dtype=np.float32)
    b = np.array([[[0.0, 1.0, 1.0], [1.0, -0.25, 1e30]]], dtype=np.float32)
    mean, rgba = oracle.frame([a, b])
    assert np.array_equal(mean[0, 0], np.array([0.0, 1.0, 2.0], dtype=np.float32))
    # 0 -> 0; 1/(1+1)*255 = 127.5 -> 127; 2/3*255 = 170.0000x or 169.9999x in f32
    assert rgba[0, 0].tolist() == [0, 127, int(np.float32(2.0) / np.float32(3.0) * np.float32(255.0)), 255]
    # NaN -> 0; negative -> 0 (saturating); 1e30/(1+1e30) = 1 -> 255
    assert rgba[0, 1].tolist() == [0, 0, 255, 255]
    # a frame without renders divides by max(0, 1)
    mean0, rgba0 = oracle.frame([a])
    assert np.array_equal(H.bits(mean0), H.bits(a))


def test_frame_sum_is_sequential_f32(oracle):
    rs = special_renders(5, 7, 4, seed=3)
    mean, _ = oracle.frame(rs)
    total = np.zeros_like(rs[0])
    with np.errstate(all="ignore"):
        for r in rs:
            total = total + r
        want = total / np.float32(4)
    assert np.array_equal(H.bits(mean), H.bits(want))


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,count", [(1, 1, 1), (3, 5, 2), (45, 61, 3), (270, 480, 7), (1080, 1920, 2)])
def test_gpu_frame_matches_the_oracle(oracle, h, w, count):
    rs = special_renders(h, w, count, seed=h * 1000 + w)
    frame = P.Frame(w, h, 0)
    assert frame.count == 0
    assert not frame.mean_image().any() and np.array_equal(frame.image()[..., 3], np.full((h, w), 255, np.uint8))
    for r in rs:
        frame.add_render(r)
    assert frame.count == count
    want_mean, want_rgba = oracle.frame(rs)
    assert np.array_equal(H.bits(frame.mean_image()), H.bits(want_mean))
    assert np.array_equal(frame.image(), want_rgba)
    frame.close()


@pytest.mark.gpu
def test_gpu_frame_of_renders(oracle):
    """Two renders of different seeds, as two workers would send them: the frame's
    mean is their average and its preview is the oracle's, byte for byte."""
    scene = P.Scene.load(H.SCENE_DATA, "./simple_scene.json")
    gpu = P.HipRenderer.init(scene, 0)
    task = H.load_task()
    task.camera.resolution = [64, 48]
    task.config = P.Config(trace_depth=8, iterations=4)
    renders = [gpu.render(task, P.RenderOpts(seed=s)) for s in (1, 2)]
    frame = P.Frame(64, 48, 0)
    for r in renders:
        # through the wire codec, as the client receives them (lib.rs:56-78)
        frame.add_render(P.RenderedImage.from_bytes(P.RenderedImage(r).to_bytes()).image)
    want_mean, want_rgba = oracle.frame(renders)
    assert np.array_equal(H.bits(frame.mean_image()), H.bits(want_mean))
    assert np.array_equal(frame.image(), want_rgba)
    assert want_rgba[..., :3].max() > 100  # not a black picture
    with pytest.raises(ValueError):
        frame.add_render(renders[0][:10])
    gpu.close()
    frame.close()


@pytest.mark.gpu
@pytest.mark.parametrize("misalign", [0, 1])
def test_gpu_frame_device_buffers(oracle, misalign):
    """Renders and outputs that stay in device memory, on a stream of the caller's;
    misalign=1 shifts every buffer off 16-byte alignment (the scalar kernels)."""
    import torch

    h, w = 33, 47
    rs = special_renders(h, w, 3, seed=9)
    frame = P.Frame(w, h, 0)
    stream = torch.cuda.Stream()
    keep = []
    with torch.cuda.stream(stream):
        for r in rs:
            t = torch.empty(h * w * 3 + 4, dtype=torch.float32, device="cuda")
            t[misalign:misalign + h * w * 3] = torch.from_numpy(r.reshape(-1)).cuda()
            keep.append(t)
            frame.add_render_device(t.data_ptr() + 4 * misalign, stream.cuda_stream)
        rgb = torch.zeros(h * w * 3 + 4, dtype=torch.float32, device="cuda")
        rgba = torch.zeros(h * w * 4 + 16, dtype=torch.uint8, device="cuda")
        frame.image_device(rgb.data_ptr() + 4 * misalign, rgba.data_ptr() + 4 * misalign, stream.cuda_stream)
    stream.synchronize()
    want_mean, want_rgba = oracle.frame(rs)
    got_mean = rgb[misalign:misalign + h * w * 3].cpu().numpy().reshape(h, w, 3)
    got_rgba = rgba[4 * misalign:4 * misalign + h * w * 4].cpu().numpy().reshape(h, w, 4)
    assert np.array_equal(H.bits(got_mean), H.bits(want_mean))
    assert np.array_equal(got_rgba, want_rgba)
    # the host getters see the same sum
    assert np.array_equal(frame.image(), want_rgba)
    frame.close()
