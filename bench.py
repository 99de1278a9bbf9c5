#!/usr/bin/env python3
"""Benchmark of the hot path: path-traced frames on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c3|c4|c5]

One step = one full render of the workload on every rank, the un-normalised sums
reduced onto rank 0 over RCCL inside the timed region, rank 0 resolving the image.
The scene is resident in HBM before timing starts.  Rank 0 prints one JSON line.

  c3 (default)  dragon stand-in (871,200 triangles), 1920x1080, 64 samples per pixel in all,
                depth 8; ranks take disjoint sample ranges of that one frame (strong scaling:
                the frame the metric names, on more GPUs).  With N > 1 the line also carries
                extra.weak_64spp_per_gpu: the same loop with 64 samples per GPU
  c4            the same scene at 3840x2160, 256 samples in all, split into sample
                ranges over the ranks (strong scaling)
  c5            115 transformed instances of the mesh (1.0e8 triangles on screen) through
                the scene graph, 3840x2160, 4 samples, split into row bands (strong scaling)

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts the N ranks
itself (python -m torch.distributed.run, one process per GPU), relays rank 0's line and
exits with the ranks' status; under torch.distributed.run it is one of the ranks.
--inproc runs the same steps in ONE process through pt_render_multi (one host thread and
one stream per device, grouped RCCL reduce): the entry point a host that is not Python
would call.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
# wave64 vector instructions per second a SIMD can issue, measured (tools/micro/valu_issue.hip, profiles/r3_valu_issue.txt):
# with two or more waves on a SIMD the simple operations (v_fma / v_add / v_sub / v_mul / v_mov / v_and / v_add_u32 /
# v_cndmask with vcc) issue every 2.2 cycles -- /opt/skills/guides/MI355X_MICROARCH.md:54,485 says 2 -- while comparisons,
# min / max / max3 / min3, byte-to-float conversions, shifts, bit-field operations and packed multiply-adds issue every
# 4.1 cycles, at any number of waves.  The peak below is the fast class at the nominal clock; the second figure prices
# the search kernel's own mix (profiles/r3_search_issue_mix.txt, tools/issue_mix.py).
SIMDS = 256 * 4
CLOCK_HZ = 2.4e9
VALU_FAST_CYCLES = 2.19
VALU_SLOW_CYCLES = 4.09
VALU_PEAK_PER_S = SIMDS * CLOCK_HZ / VALU_FAST_CYCLES
DEPTH = 8
CONFIGS = {
    "c3": {"width": 1920, "height": 1080, "spp": 64, "mode": "samples", "scaling": "strong"},
    "c4": {"width": 3840, "height": 2160, "spp": 256, "mode": "samples", "scaling": "strong"},
    "c5": {"width": 3840, "height": 2160, "spp": 4, "mode": "rows", "scaling": "strong", "instances": 115},
}
HBM_REGIME_COPIES = 24  # flattened copies of the mesh for the HBM-bound record: 2.1e7 triangles, 1.2 GB of nodes and triangles


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (profiling runs)")
    ap.add_argument("--no-hbm-regime", action="store_true", help="skip the second roofline record (flattened 2.1e7-triangle scene)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 on a box with fewer GPUs than ranks: every rank uses GPU 0, gloo instead of RCCL, the sum "
                         "staged through host memory; checks the control flow, its numbers mean nothing")
    ap.add_argument("--stub-render", action="store_true",
                    help="no GPU at all: the render is replaced by a stand-in that fills the buffer (tests of the launch and "
                         "reduce logic on CPU; implies --rehearse)")
    ap.add_argument("--inproc", action="store_true",
                    help="one process drives the N GPUs through pt_render_multi (threads, streams, grouped RCCL reduce) "
                         "instead of one process per GPU; with --rehearse all N scene handles live on GPU 0")
    ap.add_argument("--scene-dir", default=os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon"))
    args = ap.parse_args(argv)
    if args.stub_render:
        args.rehearse = True
    return args


def self_launch(args) -> int:
    """Starts the ranks as fresh processes (nothing in this process has touched the GPU:
    only the standard library is imported so far), relays rank 0's JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line, foreign = pick_result_line(proc.stdout)
    for ln in foreign:
        sys.stderr.write(ln + "\n")
    if line is not None:
        print(line, flush=True)
    if proc.returncode == 0 and line is None:
        sys.stderr.write("bench.py: the ranks printed no result line\n")
        return 1
    return proc.returncode


def pick_result_line(stdout_text):
    """Rank 0's result among everything the ranks wrote to the shared pipe: the last line that is a
    JSON object with a "metric" key.  Anything else -- banners of the collective library, a bare
    number or string (valid JSON, not an object), half a line -- is returned as foreign text and never
    raises."""
    line, foreign = None, []
    for ln in stdout_text.splitlines():
        obj = None
        try:
            obj = json.loads(ln)
        except ValueError:
            pass
        if isinstance(obj, dict) and "metric" in obj:
            line = ln
        elif ln.strip():
            foreign.append(ln)
    return line, foreign


def host_threads() -> int:
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        v = os.environ.get(var)
        if v and v.isdigit() and int(v) > 0:
            return int(v)
    return max(1, min(os.cpu_count() or 1, 16))


def cpu_legs(P, scene, task):
    """The oracle on the host cores: (a) the reference algorithm (every triangle tested)
    on a small crop, (b) its BVH mode on a larger sample."""
    from tests import oracle_binding as ob  # test infrastructure: checker / baseline only

    oracle = ob.load()
    osc = oracle.scene(scene.desc_pointer)
    threads = host_threads()
    t = P.RenderTask(scene=task.scene, config=P.Config(DEPTH, 1), camera=P.Camera(**{**task.camera.__dict__}))
    t.camera.resolution = [480, 270]
    _, st = osc.render(t, P.RenderOpts(seed=0), accel=ob.ACCEL_BVH, math=ob.MATH_LIBM, threads=threads)
    bvh = {"value": st.rays / st.seconds / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"oracle in BVH mode (its own BVH2), same scene and camera at 480x270, 1 sample per pixel, depth 8: "
                     f"{int(st.rays)} rays in {st.seconds * 1e3:.0f} ms"}
    t.camera.resolution = [32, 18]
    _, st2 = osc.render(t, P.RenderOpts(seed=0), accel=ob.ACCEL_LINEAR, math=ob.MATH_LIBM, threads=threads)
    ref = {"value": st2.rays / st2.seconds / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"oracle running the reference algorithm (linear scan of all 871,200 triangles per ray), same scene "
                     f"and camera at 32x18, 1 sample per pixel, depth 8: {int(st2.rays)} rays in {st2.seconds:.1f} s"}
    osc.close()
    return ref, bvh


def search_kernel_record(P, renderer, task, opts_base, accum_ptr, stream):
    """The dominant kernel measured live: one render with events around every search_kernel
    launch (durations) and one with the kernel counting the records it fetches (bytes).
    Algorithmic bytes of a launch, as the kernel's data layout has them: one 64-B node record
    per node visited (four child boxes), one 48-B triangle record per triangle tested, and
    per search the 48-B job read and the 16-B result written."""
    o = P.RenderOpts(**opts_base.__dict__)
    o.profile = True
    prof = renderer.render_accumulate_device(task, o, accum_ptr, stream, want_stats=True)
    o = P.RenderOpts(**opts_base.__dict__)
    o.count_visits = True
    o.single_stream = True
    cnt = renderer.render_accumulate_device(task, o, accum_ptr, stream, want_stats=True)
    launches = int(prof.search_kernel_launches)
    if launches == 0 or cnt.searches == 0:
        return None
    launch_ms = prof.search_kernel_ms / launches
    bytes_per_launch = (64.0 * cnt.node_visits + 48.0 * cnt.triangle_tests + 64.0 * cnt.searches) / launches
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
    return {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": None, "kernel": "search_kernel", "bytes_per_launch": bytes_per_launch, "launch_ms": launch_ms,
        "launches": launches, "searches_per_launch": cnt.searches / launches,
        "node_visits_per_search": cnt.node_visits / cnt.searches, "triangle_tests_per_search": cnt.triangle_tests / cnt.searches,
        "bytes_per_search": bytes_per_launch * launches / cnt.searches,
        "searches_per_ray": cnt.searches / cnt.rays, "search_kernel_share_of_render": prof.search_kernel_ms / prof.render_ms,
        "searches_per_s_in_kernel": cnt.searches / (prof.search_kernel_ms * 1e-3),
        "bvh_bytes": int(cnt.bvh_bytes), "render_ms_profiled": prof.render_ms,
        # of a wave's 64 lanes, how many are at work in a node step / a triangle round (each lane runs its own search)
        "lanes_per_node_step": cnt.node_visits / max(cnt.wave_node_steps, 1),
        "lanes_per_triangle_round": cnt.triangle_tests / max(cnt.wave_leaf_rounds, 1),
    }


def recorded(name, fingerprint):
    """A measurement that cannot be taken inside this process (rocprofv3 PMC passes), as recorded
    under profiles/ by tools/collect_profiles.sh -- but only when it was taken with the kernels this
    process has loaded: the record carries the fingerprint of the library's device code, and one
    taken with other kernels is not replayed.  Returns (record or None, note or None)."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None, f"profiles/{name} is absent"
    if rec.get("kernel_fingerprint") != fingerprint:
        return None, (f"profiles/{name} was recorded with other kernels (device code {rec.get('kernel_fingerprint')}, "
                      f"loaded {fingerprint}): run tools/collect_profiles.sh again")
    return rec, None


def hbm_regime_record(P, N, synth, scene_dir, device, torch, fingerprint):
    """The same kernel where it is bound by memory: HBM_REGIME_COPIES transformed copies of the
    mesh flattened into ONE mesh (2.1e7 world-space triangles, 1.2 GB of nodes and triangles,
    five times the Infinity Cache), 3840x2160, 2 samples per pixel."""
    import ctypes as C

    import numpy as np

    with open(os.path.join(scene_dir, "flattened.json"), "w") as f:
        json.dump(synth.flattened_scene_json(), f)
    with open(os.path.join(scene_dir, "dragon_task.json")) as f:
        task = P.RenderTask.from_json(f.read())
    task.scene = "./flattened.json"
    task.camera.resolution = [3840, 2160]
    task.camera.position = [0.0, 0.0, 330.0]
    task.camera.focal_length = 330.0
    task.config = P.Config(trace_depth=DEPTH, iterations=2)
    scene = P.Scene.load(scene_dir, task.scene)
    base = scene.desc
    n1 = int(base.meshes[0].triangle_count)
    one = np.ctypeslib.as_array(base.meshes[0].positions, shape=(n1, 9))
    positions = synth.flattened_copies(one, HBM_REGIME_COPIES, seed=5)
    mesh = N.PtMeshDesc()
    mesh.triangle_count = positions.shape[0]
    mesh.positions = positions.ctypes.data_as(C.POINTER(C.c_float))
    big = N.PtSceneDesc()
    for name, _ in N.PtSceneDesc._fields_:
        setattr(big, name, getattr(base, name))
    big.meshes = C.pointer(mesh)
    big.mesh_count = 1
    t0 = time.time()
    gpu = P.HipRenderer.from_desc(C.pointer(big), device)
    setup_s = time.time() - t0
    accum = torch.zeros(3840 * 2160 * 3, dtype=torch.float32, device=f"cuda:{device}")
    stream = torch.cuda.current_stream().cuda_stream
    gpu.render_accumulate_device(task, P.RenderOpts(seed=0), accum.data_ptr(), stream, want_stats=True)  # warm-up
    rec = search_kernel_record(P, gpu, task, P.RenderOpts(seed=0), accum.data_ptr(), stream)
    timed = gpu.render_accumulate_device(task, P.RenderOpts(seed=0), accum.data_ptr(), stream, want_stats=True)
    gpu.close()
    if rec is None:
        return None
    rec["workload"] = (f"{HBM_REGIME_COPIES} transformed copies of the mesh flattened into one mesh of {positions.shape[0]} "
                       f"world-space triangles, 3840x2160, 2 samples per pixel, depth {DEPTH}")
    rec["mrays_per_s"] = timed.rays / timed.render_ms / 1e3
    rec["scene_setup_s"] = setup_s
    t, note = recorded("hbm_traffic_flattened.json", fingerprint)
    rec["traffic"] = t["bytes_per_frame"] / rec["launches"] if t and t.get("bytes_per_frame") else None
    if note:
        rec["traffic_note"] = note
    return rec


class StubRenderer:
    """Stands in for the GPU in tests of the launch / shard / reduce logic: every sample
    adds 1 to every pixel of the rows it covers."""

    def __init__(self, width, height):
        self.width, self.height = width, height

    def render_accumulate(self, task, opts, buf):
        import types

        x0, y0, w, h = opts.tile if opts.tile else (0, 0, self.width, self.height)
        view = buf.view(self.height, self.width, 3)
        view[y0:y0 + h, x0:x0 + w, :] += float(task.config.iterations)
        paths = task.config.iterations * w * h
        return types.SimpleNamespace(rays=paths, paths=paths, searches=0, render_ms=0.0)


def issue_record(sq, roofline):
    """Vector-instruction issue of the dominant kernel against what the SIMDs can issue (measured peaks above)."""
    k = sq["search_kernel"]
    insts = k["valu_insts_per_frame"] / roofline["launches"]
    achieved = insts / (roofline["launch_ms"] * 1e-3)
    rec = {"bound": "valu", "achieved": achieved, "peak": VALU_PEAK_PER_S, "unit": "wave64 VALU instructions/s",
           "frac": achieved / VALU_PEAK_PER_S, "active_lane_share": k.get("active_lane_share"),
           "source": "profiles/sq_counters.json", "peak_source": "profiles/r3_valu_issue.txt"}
    slow = k.get("slow_class_share")  # static share of the kernel's vector instructions that issue every 4.1 cycles
    if slow is not None:
        cycles = (1.0 - slow) * VALU_FAST_CYCLES + slow * VALU_SLOW_CYCLES
        rec["mix_peak"] = SIMDS * CLOCK_HZ / cycles
        rec["mix_frac"] = achieved / rec["mix_peak"]
        rec["slow_class_share"] = slow
    return rec


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and not args.inproc and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    multi = world > 1  # one process per GPU
    if args.inproc:
        if multi:
            raise SystemExit("--inproc is one process: do not start it under torch.distributed.run")
        if args.stub_render:
            raise SystemExit("--inproc drives real devices: it has no stand-in render")
    elif args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    shares = args.gpus  # GPUs the frame is cut over, whichever way they are driven

    import numpy as np  # noqa: F401
    import torch  # device buffers, streams and the RCCL reduce; not on the compute path

    import pathtracing_amd as P  # loads the HIP library; fails loudly if it is missing
    from pathtracing_amd import _native as N
    from pathtracing_amd import synth
    from pathtracing_amd.distributed import render_sharded, shard_opts

    fingerprint = N.kernel_fingerprint()
    cfg = dict(CONFIGS[args.config])
    if args.stub_render:
        cfg.update(width=64, height=40, spp=8)
    width, height, spp = cfg["width"], cfg["height"], cfg["spp"]

    if multi:
        import torch.distributed as dist

        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, "the process group does not have one rank per requested GPU"
    device = local_rank if multi and not args.rehearse else 0
    tdev = "cpu" if args.stub_render else f"cuda:{device}"
    if not args.stub_render:
        torch.cuda.set_device(device)
    my_devices = [device]
    if args.inproc:
        my_devices = [0] * shares if args.rehearse else list(range(shares))
        if not args.rehearse and torch.cuda.device_count() < shares:
            raise SystemExit(f"--inproc --gpus {shares}: only {torch.cuda.device_count()} devices visible (--rehearse puts all on GPU 0)")

    # ---- the scene: rank 0 of the node writes the files, the others wait; every rank uploads its own copy
    setup_s, scene, renderer, info, renderers = 0.0, None, None, None, []
    if args.stub_render:
        task = P.RenderTask.from_json(json.dumps(synth.dragon_render_task()))
        renderer = StubRenderer(width, height)
    else:
        if local_rank == 0:
            synth.write_dragon_scene(args.scene_dir, os.path.join(ROOT, "scene_data"))
            if args.config == "c5":
                with open(os.path.join(args.scene_dir, "instanced.json"), "w") as f:
                    json.dump(synth.instanced_dragon_scene_json(cfg["instances"], seed=3), f)
        if multi:
            dist.barrier()
        with open(os.path.join(args.scene_dir, "dragon_task.json")) as f:
            task = P.RenderTask.from_json(f.read())
        if args.config == "c5":
            task.scene = "./instanced.json"
            task.camera.position = [0.0, 10.0, 110.0]
            task.camera.focal_length = 110.0
        scene = P.Scene.load(args.scene_dir, task.scene)
        t0 = time.time()
        # BVH build (or .tree adoption) + upload: outside the timed region
        renderers = [P.HipRenderer.init(scene, d) for d in my_devices]
        renderer = renderers[0]
        setup_s = time.time() - t0
        info = renderer.info
    task.camera.resolution = [width, height]

    accum = torch.zeros(height * width * 3, dtype=torch.float32, device=tdev)
    out = torch.empty_like(accum)
    stream = 0 if args.stub_render else torch.cuda.current_stream().cuda_stream
    last = {}
    reduce_spans = []  # (start, end): CUDA events, or host seconds

    def render(sub, opts, buf):
        if args.stub_render:
            last["stats"] = renderer.render_accumulate(sub, opts, buf)
        else:
            last["stats"] = renderer.render_accumulate_device(sub, opts, buf.data_ptr(), stream, want_stats=True)

    def mark_reduce(which):
        if args.stub_render or args.rehearse:
            stamp = time.perf_counter()
        else:
            stamp = torch.cuda.Event(enable_timing=True)
            stamp.record()  # the reduce is ordered against the current stream on both sides
        if which == "start":
            reduce_spans.append([stamp, None])
        else:
            reduce_spans[-1][1] = stamp

    def step(frame, total_spp):
        last["stats"] = None
        if args.inproc:
            # one call: every device renders its share, grouped reduce onto the first device, get_image into `out`
            _, st = P.render_multi(renderers, frame, P.RenderOpts(seed=0), cfg["mode"], out_ptr=out.data_ptr(),
                                   width_height=(width, height))
            reduce_spans.append([0.0, st.reduce_ms * 1e-3])
            return st
        accum.zero_()
        # this rank's share, then the sum of all ranks' buffers onto rank 0 (RCCL reduce), then get_image on rank 0
        render_sharded(render, frame, accum, mode=cfg["mode"], seed=0, stage_on_host=args.rehearse and not args.stub_render,
                       on_reduce=mark_reduce)
        if rank == 0:
            if args.stub_render:
                out.copy_(accum * (1.0 / total_spp))
            else:
                N.check(N.lib.pt_resolve_device(accum.data_ptr(), width, height, total_spp, out.data_ptr(), stream))
        return last["stats"]

    def fence():
        if multi:
            dist.barrier()
        if not args.stub_render:
            for d in set(my_devices):
                torch.cuda.synchronize(d)

    def timed_loop(total_spp):
        """W untimed steps, then K steps between fences; the slowest rank's time, every rank's rays."""
        frame = P.RenderTask(scene=task.scene, scene_md5=task.scene_md5, camera=task.camera,
                             config=P.Config(trace_depth=DEPTH, iterations=total_spp))
        for _ in range(args.warmup):
            step(frame, total_spp)
        fence()
        del reduce_spans[:]
        t0 = time.perf_counter()
        rays = searches = 0
        for _ in range(args.steps):
            st = step(frame, total_spp)
            if st is not None:
                rays += st.rays
                searches += st.searches
        fence()
        elapsed = time.perf_counter() - t0
        reduce_s = 0.0
        for a, b in reduce_spans:
            reduce_s += a.elapsed_time(b) * 1e-3 if hasattr(a, "elapsed_time") else b - a
        if multi:
            sdev = "cpu" if args.rehearse else f"cuda:{device}"
            tmax = torch.tensor([elapsed, reduce_s], dtype=torch.float64, device=sdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed, reduce_s = float(tmax[0].item()), float(tmax[1].item())
            total = torch.tensor([rays, searches], dtype=torch.int64, device=sdev)
            dist.all_reduce(total, op=dist.ReduceOp.SUM)
            rays, searches = int(total[0].item()), int(total[1].item())
        return {"frame": frame, "elapsed": elapsed, "rays": rays, "searches": searches,
                "reduce_ms": reduce_s / max(args.steps, 1) * 1e3 if shares > 1 else 0.0,
                "frame_mean": float(out.mean().item()) if rank == 0 else None}

    # the frame the metric names: `spp` samples in all, cut over the GPUs
    head = timed_loop(spp)
    # and, for the default workload on several GPUs, the same loop with `spp` samples on every GPU
    weak = timed_loop(spp * shares) if shares > 1 and args.config == "c3" else None
    devices = list(my_devices)
    if multi:
        gathered = [None] * world
        dist.all_gather_object(gathered, device)
        devices = gathered
    if shares > 1 and not args.rehearse and len(set(devices)) != shares:
        raise SystemExit(f"{shares} shares on devices {devices}: not one GPU each")

    result = None
    if rank == 0:
        frame, elapsed, rays, searches = head["frame"], head["elapsed"], head["rays"], head["searches"]
        roofline = hbm_regime = ref = bvh = None
        if not args.stub_render:
            # untimed: this rank's share once more (no collective: the other ranks have left the loop), with
            # events around every search_kernel launch, and once with the kernel counting what it fetches
            sub, o = shard_opts(frame, shares, 0, cfg["mode"], seed=0)
            roofline = search_kernel_record(P, renderer, sub, o, accum.data_ptr(), stream)
            torch.cuda.synchronize()
            if roofline is not None:
                roofline["kernel_fingerprint"] = fingerprint
                whole_frame = args.config == "c3" and shares == 1  # what the recorded counter passes rendered
                # recorded per frame (rocprofv3 PMC passes over the same frame), divided by this run's launches per frame
                t, note = recorded("hbm_traffic.json", fingerprint) if whole_frame else (None, None)
                roofline["traffic"] = t["bytes_per_frame"] / roofline["launches"] if t and t.get("bytes_per_frame") else None
                if note:
                    roofline["traffic_note"] = note
                # the working set (nodes + leaf triangles) against the 256 MiB Infinity Cache: below it the records are
                # re-read from cache and the figure above is a rate of cache traffic, bounded by the HBM peak only by name
                roofline["cache_resident"] = roofline["bvh_bytes"] < 256 * 2**20
                sq, note = recorded("sq_counters.json", fingerprint) if whole_frame else (None, None)
                if sq and sq.get("search_kernel") and sq["search_kernel"].get("valu_insts_per_frame"):
                    roofline["issue"] = issue_record(sq, roofline)
                elif note:
                    roofline["issue"] = None
                    roofline["issue_note"] = note
            if shares == 1 and args.config == "c3":
                if not args.no_hbm_regime:
                    hbm_regime = hbm_regime_record(P, N, synth, args.scene_dir, device, torch, fingerprint)
                if not args.no_cpu_baseline:
                    ref, bvh = cpu_legs(P, scene, task)
        workloads = {
            "c3": f"dragon stand-in (871,200 triangles, torus-knot tube) in a textured box, {width}x{height}, {spp} samples per "
                  f"pixel in all, trace depth {DEPTH}",
            "c4": f"dragon stand-in (871,200 triangles) in a textured box, {width}x{height}, {spp} samples per pixel in all, "
                  f"trace depth {DEPTH}",
            "c5": f"{cfg.get('instances')} transformed instances of the 871,200-triangle mesh through the scene graph "
                  f"(1.0e8 triangles on screen), {width}x{height}, {spp} samples per pixel in all, trace depth {DEPTH}",
        }
        split = {"samples": "sample range per GPU", "rows": "band of 8-pixel rows per GPU"}[cfg["mode"]]
        how = ("one process, pt_render_multi: a host thread and a stream per device, grouped RCCL reduce(sum) onto the first"
               if args.inproc else "one process per GPU, RCCL reduce(sum) of the accumulation buffers onto rank 0")
        rehearsal = "REHEARSAL (every share on GPU 0, no RCCL): not a measurement; " if args.rehearse else ""
        result = {
            "metric": "Mrays/s dragon 1080p 64spp" if args.config == "c3" else f"Mrays/s {args.config}",
            "value": rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": shares, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("STAND-IN RENDER (no GPU): launch and reduce logic only; " if args.stub_render else "") + workloads[args.config],
                "sharding": f"{rehearsal}{split}; {how}" if shares > 1 else "single GPU",
                "ranks": shares, "devices": devices, "processes": world,
                "rays_per_step": rays // max(args.steps, 1), "searches_per_step": searches // max(args.steps, 1),
                "searches_per_s": searches / elapsed, "scene_setup_s": setup_s,
                "trees_adopted": int(info.trees_adopted) if info else None, "bvh_build_s": info.build_ms / 1e3 if info else None,
                "frame_mean": head["frame_mean"], "reduce_ms": head["reduce_ms"],
            },
            "roofline": roofline, "roofline_hbm_regime": hbm_regime, "cpu_baseline": ref, "cpu_baseline_bvh": bvh,
        }
        if weak is not None:
            result["extra"] = {"weak_64spp_per_gpu": {
                "value": weak["rays"] / weak["elapsed"] / 1e6, "unit": "Mrays/s", "scaling": "weak",
                "ms_per_step": weak["elapsed"] / args.steps * 1e3, "samples_per_pixel_in_all": spp * shares,
                "rays_per_step": weak["rays"] // max(args.steps, 1), "reduce_ms": weak["reduce_ms"], "frame_mean": weak["frame_mean"]}}
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if result is not None:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
