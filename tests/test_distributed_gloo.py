"""The N>1 path on CPU: two processes, gloo backend.  Each rank renders its share
with the CPU oracle standing in for the GPU (same accumulate-into-buffer contract
as pt_render_accumulate_device), the buffers are reduced onto rank 0 and compared
with a single-process render."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pathtracing_amd as P
from pathtracing_amd import distributed as D
from tests import helpers as H
from tests import oracle_binding as ob


def test_sample_ranges_partition():
    for iters in (0, 1, 7, 64, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [D.sample_range(iters, world, r) for r in range(world)]
            assert sum(c for _, c in spans) == iters
            pos = 0
            for first, count in spans:
                assert first == pos
                pos += count
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_row_bands_partition():
    for height in (1, 8, 45, 1080, 2160):
        for world in (1, 2, 3, 8):
            bands = [D.row_band(height, world, r) for r in range(world)]
            assert sum(n for _, n in bands) == height
            pos = 0
            for y0, n in bands:
                if n:
                    assert y0 == pos and y0 % 8 == 0
                    pos += n
    with pytest.raises(ValueError):
        D.row_band(10, 2, 2)


def _oracle_accumulate(osc, width, height):
    def fn(task, opts, accum):
        img, _ = osc.render(task, opts, accel=ob.ACCEL_LINEAR, math=ob.MATH_PORTABLE, threads=2)
        # undo get_image: the oracle returns sum/iterations with rows flipped
        sums = img[::-1].astype(np.float64) * task.config.iterations
        accum += torch.from_numpy(sums.astype(np.float32).reshape(-1))
    return fn


def _worker(rank, world, port, mode, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        task = H.load_task()
        task.camera.resolution = [40, 28]
        task.config.iterations = 5
        scene = P.Scene.load(H.SCENE_DATA, task.scene)
        osc = ob.load().scene(scene.desc_pointer)
        accum = torch.zeros(40 * 28 * 3, dtype=torch.float32)
        D.render_sharded(_oracle_accumulate(osc, 40, 28), task, accum, mode=mode, seed=9)
        if rank == 0:
            np.save(out_path, D.resolve_host(accum.numpy(), 40, 28, 5))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["samples", "rows"])
def test_two_ranks_match_single_process(tmp_path, mode):
    out = str(tmp_path / f"{mode}.npy")
    mp.spawn(_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    got = np.load(out)
    task = H.load_task()
    task.camera.resolution = [40, 28]
    task.config.iterations = 5
    scene = P.Scene.load(H.SCENE_DATA, task.scene)
    want, _ = ob.load().scene(scene.desc_pointer).render(task, P.RenderOpts(seed=9), math=ob.MATH_PORTABLE)
    if mode == "rows":
        # every pixel is rendered by exactly one rank with all its samples: only the
        # scale-by-iterations round trip of this test's stand-in separates the two
        assert np.allclose(got, want, rtol=2e-6, atol=1e-7)
    else:
        # sample ranges change the f32 summation order
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6)
    assert got.mean() > 0.05
