"""Sharding one render task over the GPUs of a node (one process per GPU).

The reference combines workers on the host: the client sends the same task to
every worker and averages the returned frames (client/src/worker_pool.rs:221-236,
client/src/frame.rs:60-84).  Inside a node the same combination is one
collective: every rank renders its share into a camera-space buffer of
un-normalised sums, the buffers are summed onto rank 0 (RCCL over xGMI with the
``nccl`` backend, ``gloo`` on CPU for tests) and rank 0 scales by
``1 / iterations`` and flips the rows like ``get_image``
(cpu_renderer/mod.rs:143-175).

Two ways to cut the work, both free of any exchange while rendering:
  * sample range: rank r renders samples [offset_r, offset_r + count_r) of every
    pixel; the summed image differs from a single-GPU render only in f32
    summation order;
  * row bands: rank r renders every sample of its band of 8-pixel rows; other
    pixels stay zero, so the same sum reassembles the frame bit for bit.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

from .api import RenderOpts, RenderTask


def sample_range(iterations: int, world: int, rank: int) -> Tuple[int, int]:
    """(first sample, number of samples) of `rank`; ranges are contiguous, disjoint
    and cover [0, iterations); the first `iterations % world` ranks get one more."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    base, extra = divmod(iterations, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def row_band(height: int, world: int, rank: int) -> Tuple[int, int]:
    """(first row, number of rows) of `rank` in camera space; bands are multiples of
    8 rows (the kernels' pixel-block height) except the last one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    blocks = (height + 7) // 8
    base, extra = divmod(blocks, world)
    nblk = base + (1 if rank < extra else 0)
    first_blk = rank * base + min(rank, extra)
    y0 = first_blk * 8
    rows = max(0, min(height, (first_blk + nblk) * 8) - y0)
    return y0, rows


def shard_opts(task: RenderTask, world: int, rank: int, mode: str, seed: int = 0,
               base: Optional[RenderOpts] = None) -> Tuple[RenderTask, Optional[RenderOpts]]:
    """The (task, opts) `rank` should render, or (task, None) when it has nothing to do."""
    o = RenderOpts(seed=seed) if base is None else RenderOpts(**base.__dict__)
    o.seed = seed
    if mode == "samples":
        first, count = sample_range(task.config.iterations, world, rank)
        if count == 0:
            return task, None
        sub = RenderTask(scene=task.scene, scene_md5=task.scene_md5, camera=task.camera,
                         config=type(task.config)(trace_depth=task.config.trace_depth, iterations=count))
        o.sample_offset = o.sample_offset + first
        return sub, o
    if mode == "rows":
        y0, rows = row_band(task.camera.resolution[1], world, rank)
        if rows == 0:
            return task, None
        o.tile = (0, y0, task.camera.resolution[0], rows)
        return task, o
    raise ValueError(f"unknown sharding mode {mode!r}")


def render_sharded(render_accumulate: Callable[[RenderTask, RenderOpts, "object"], None], task: RenderTask,
                   accum, mode: str = "samples", seed: int = 0, group=None, stage_on_host: bool = False,
                   on_reduce: Optional[Callable[[str], None]] = None):
    """Renders this rank's share into `accum` (a zeroed torch tensor of W*H*3 f32 on
    the rank's device, camera-space layout) and sums all ranks' buffers onto rank 0.
    `render_accumulate(task, opts, accum)` must add the samples it renders into
    `accum`.  Returns `accum` (complete on rank 0 only).  `stage_on_host` sums through
    a host copy, for process groups whose backend cannot reduce device tensors (gloo).
    `on_reduce("start")` / `on_reduce("end")` are called around the sum (for timing it)."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sub, opts = shard_opts(task, world, rank, mode, seed)
    if opts is not None:
        render_accumulate(sub, opts, accum)
    if world > 1:
        if on_reduce is not None:
            on_reduce("start")
        if stage_on_host:
            host = accum.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM, group=group)
            accum.copy_(host)
        else:
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM, group=group)
        if on_reduce is not None:
            on_reduce("end")
    return accum


def resolve_host(accum, width: int, height: int, iterations: int):
    """get_image on the host for CPU tests: scale by 1/iterations (as an f32
    reciprocal, work_group.rs:67-69) and flip the rows."""
    import numpy as np

    a = np.asarray(accum, dtype=np.float32).reshape(height, width, 3)
    scale = np.float32(1.0) / np.float32(iterations)
    return (a * scale)[::-1].copy()
