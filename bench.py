#!/usr/bin/env python3
"""Benchmark of the hot path: path-traced frames of the dragon scene on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one full render of the workload (1920x1080, 64 samples per pixel,
trace depth 8, ~871k-triangle dragon stand-in) on every rank; ranks render
disjoint sample ranges of the same frame and the un-normalised sums are reduced
to rank 0 over RCCL inside the timed region (weak scaling: per-GPU work fixed).
The scene is resident in HBM before timing starts.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import pathtracing_amd as P  # noqa: E402  (loads the HIP library; fails loudly if it is missing)
from pathtracing_amd import synth  # noqa: E402
from pathtracing_amd.distributed import render_sharded, shard_opts  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
WIDTH, HEIGHT, SPP, DEPTH = 1920, 1080, 64, 8


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (profiling runs)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 on a box with fewer GPUs than ranks: every rank uses GPU 0, gloo instead of RCCL, the sum "
                         "staged through host memory; checks the control flow, its numbers mean nothing")
    ap.add_argument("--scene-dir", default=os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon"))
    return ap.parse_args()


def host_threads() -> int:
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        v = os.environ.get(var)
        if v and v.isdigit() and int(v) > 0:
            return int(v)
    return max(1, min(os.cpu_count() or 1, 16))


def cpu_legs(scene: P.Scene, task: P.RenderTask):
    """The oracle on the host cores: (a) the reference algorithm (every triangle
    tested) on a small crop, (b) its BVH mode on a larger sample, which also
    measures the per-ray statistics of the algorithmic byte count."""
    from tests import oracle_binding as ob  # test infrastructure: checker / baseline only

    oracle = ob.load()
    osc = oracle.scene(scene.desc_pointer)
    threads = host_threads()
    # (b) BVH mode, 480x270, 1 sample
    t = P.RenderTask(scene=task.scene, config=P.Config(DEPTH, 1), camera=P.Camera(**{**task.camera.__dict__}))
    t.camera.resolution = [480, 270]
    _, st = osc.render(t, P.RenderOpts(seed=0), accel=ob.ACCEL_BVH, math=ob.MATH_LIBM, threads=threads)
    stats = {
        "n_box": st.box_tests / st.rays,
        "n_tri": st.tri_tests / st.rays,
        "hit_fraction": st.hits / st.rays,
        "texture_fraction": st.tex_fetches / st.rays,
        "mean_path_length": st.rays / st.paths,
        "sample_rays": int(st.rays),
    }
    bvh = {"value": st.rays / st.seconds / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": "oracle in BVH mode, same scene and camera at 480x270, 1 sample per pixel, depth 8"}
    # (a) linear scan, 32x18, 1 sample
    t.camera.resolution = [32, 18]
    _, st2 = osc.render(t, P.RenderOpts(seed=0), accel=ob.ACCEL_LINEAR, math=ob.MATH_LIBM, threads=threads)
    ref = {"value": st2.rays / st2.seconds / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"oracle running the reference algorithm (linear scan of all 871,200 triangles per ray), same scene "
                     f"and camera at 32x18, 1 sample per pixel, depth 8: {int(st2.rays)} rays in {st2.seconds:.1f} s"}
    osc.close()
    return ref, bvh, stats


def bytes_per_ray(s) -> float:
    """Algorithmic HBM bytes of one path segment: 32 B per box test (bounds +
    topology), 36 B per triangle test (nine f32 positions), 60 B of shading data
    per hit, 4 B per texel, 128 B of queue traffic (48 B ray record and 16 B of
    hit state, each written and read once), and the 12 B accumulator
    read-modify-write per path spread over its segments."""
    return (32.0 * s["n_box"] + 36.0 * s["n_tri"] + 60.0 * s["hit_fraction"] + 4.0 * s["texture_fraction"] + 128.0
            + 24.0 / s["mean_path_length"])


def search_bytes_per_ray(s, searches_per_ray: float) -> float:
    """The share of bytes_per_ray() that the BVH search kernel moves: the box and
    triangle tests of a ray, and per search the 48 B request it reads and the
    16 B result it writes."""
    return 32.0 * s["n_box"] + 36.0 * s["n_tri"] + 64.0 * searches_per_ray


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    multi = world > 1
    if multi and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not multi and args.gpus != 1:
        raise SystemExit("launch N>1 through torch.distributed.run (one process per GPU)")

    import torch  # device buffers, streams and the RCCL reduce; not on the compute path

    if multi:
        import torch.distributed as dist

        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = local_rank if multi and not args.rehearse else 0
    torch.cuda.set_device(device)

    # scene files: rank 0 of the node writes them, the others wait
    if local_rank == 0:
        task_path = synth.write_dragon_scene(args.scene_dir, os.path.join(ROOT, "scene_data"))
    if multi:
        dist.barrier()
    task_path = os.path.join(args.scene_dir, "dragon_task.json")
    with open(task_path) as f:
        task = P.RenderTask.from_json(f.read())
    task.camera.resolution = [WIDTH, HEIGHT]
    task.config = P.Config(trace_depth=DEPTH, iterations=SPP)
    scene = P.Scene.load(args.scene_dir, task.scene)
    t0 = time.time()
    renderer = P.HipRenderer.init(scene, device)  # BVH build + upload: outside the timed region
    setup_s = time.time() - t0

    accum = torch.zeros(HEIGHT * WIDTH * 3, dtype=torch.float32, device=f"cuda:{device}")
    out = torch.empty_like(accum)
    stream = torch.cuda.current_stream().cuda_stream
    # weak scaling: the frame gets SPP samples per GPU; rank r renders samples [r*SPP, (r+1)*SPP)
    frame = P.RenderTask(scene=task.scene, scene_md5=task.scene_md5, camera=task.camera,
                         config=P.Config(trace_depth=DEPTH, iterations=SPP * world))
    last = {}

    def step():
        accum.zero_()

        def render(sub, opts, buf):
            last["stats"] = renderer.render_accumulate_device(sub, opts, buf.data_ptr(), stream, want_stats=True)

        # renders, then RCCL reduce(sum) onto rank 0
        render_sharded(render, frame, accum, mode="samples", seed=0, stage_on_host=args.rehearse)
        if rank == 0:
            P._native.check(P._native.lib.pt_resolve_device(accum.data_ptr(), WIDTH, HEIGHT, SPP * world, out.data_ptr(), stream))
        return last["stats"]

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    rays = 0
    for _ in range(args.steps):
        rays += step().rays
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        scalar_device = "cpu" if args.rehearse else f"cuda:{device}"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=scalar_device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        total = torch.tensor([rays], dtype=torch.int64, device=scalar_device)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        rays = int(total.item())

    result = None
    if rank == 0:
        frame_mean = float(out.mean().item())
        # one more, untimed, render of this rank's share (no collective: the other ranks
        # have left the loop) with events around every search_kernel launch
        sub, o = shard_opts(frame, world, 0, "samples", seed=0)
        o.profile = True
        prof = renderer.render_accumulate_device(sub, o, accum.data_ptr(), stream, want_stats=True)
        torch.cuda.synchronize()
        if args.no_cpu_baseline:
            ref, bvh, stats = None, None, None
        elif multi:
            # the CPU legs run at N = 1 only; the per-ray statistics of the workload do not
            # depend on N and are taken from the record of an N = 1 run
            ref, bvh, stats = None, None, recorded_ray_statistics()
        else:
            ref, bvh, stats = cpu_legs(scene, task)
        roofline = None
        if stats is not None and prof.search_kernel_launches:
            # dominant kernel: search_kernel (one launch per path segment and mesh node)
            b_search = search_bytes_per_ray(stats, prof.searches / prof.rays)
            launch_ms = prof.search_kernel_ms / prof.search_kernel_launches
            rays_per_launch = prof.rays / prof.search_kernel_launches
            achieved = b_search * rays_per_launch / (launch_ms * 1e-3) / 1e9
            b_ray = bytes_per_ray(stats)
            roofline = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(), "kernel": "search_kernel",
                "bytes_per_ray": b_search, "rays_per_launch": rays_per_launch, "launch_ms": launch_ms,
                "launches": int(prof.search_kernel_launches), "searches_per_ray": prof.searches / prof.rays,
                "ray_statistics": stats,
                # all kernels of the render against all algorithmic bytes of a path segment
                "whole_render": {"bytes_per_ray": b_ray, "achieved": b_ray * prof.rays / (prof.render_ms * 1e-3) / 1e9,
                                 "frac": b_ray * prof.rays / (prof.render_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "render_ms": prof.render_ms},
            }
        result = {
            "metric": "Mrays/s dragon 1080p 64spp", "value": rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"dragon stand-in (871,200 triangles, torus-knot tube) in a textured box, {WIDTH}x{HEIGHT}, "
                            f"{SPP} samples per pixel per GPU, trace depth {DEPTH}",
                "sharding": ("REHEARSAL on one GPU (gloo, host-staged sum): not a measurement" if args.rehearse else
                             "sample range per GPU, RCCL reduce(sum) of the accumulation buffers onto rank 0") if multi else "single GPU",
                "rays_per_step": rays // args.steps, "scene_setup_s": setup_s, "frame_mean": frame_mean,
            },
            "roofline": roofline, "cpu_baseline": ref, "cpu_baseline_bvh": bvh,
        }
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if result is not None:
        print(json.dumps(result), flush=True)


def recorded_ray_statistics():
    try:
        with open(os.path.join(ROOT, "profiles", "ray_statistics.json")) as f:
            return json.load(f).get("ray_statistics")
    except (OSError, ValueError):
        return None


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes, when a
    measurement for this workload has been recorded under profiles/."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get("bytes_per_launch")
    except (OSError, ValueError):
        return None


if __name__ == "__main__":
    main()
