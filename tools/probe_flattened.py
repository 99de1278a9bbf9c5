"""Times the flattened-instances workload: `copies` transformed copies of the
871k-triangle mesh as ONE mesh of world-space triangles (115 copies = 1.0e8
triangles), rendered at the given size.

python3 tools/probe_flattened.py [copies] [width] [height] [iterations] [check_rays]

Prints the build / upload time, device bytes of the search structure, a
closest-hit comparison of the BVH search with the linear scan on the GPU for
`check_rays` random rays, and the render rate with the search kernel's launch
times.
"""
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as P  # noqa: E402
from pathtracing_amd import _native as N  # noqa: E402
from pathtracing_amd import synth  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 115
width = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
height = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 4
check_rays = int(sys.argv[5]) if len(sys.argv) > 5 else 2048
T0 = time.time()


def say(msg):
    print(f"[{time.time() - T0:7.1f}s] {msg}", flush=True)


out = os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon")
tp = synth.write_dragon_scene(out, os.path.join(ROOT, "scene_data"))
with open(os.path.join(out, "flattened.json"), "w") as f:
    json.dump(synth.flattened_scene_json(), f)
task = P.RenderTask.from_json(open(tp).read())
task.scene = "./flattened.json"
task.camera.resolution = [width, height]
task.camera.position = [0.0, 0.0, 330.0]
task.camera.focal_length = 330.0
task.config.iterations = iters
scene = P.Scene.load(out, task.scene)
base = scene.desc
n1 = int(base.meshes[0].triangle_count)
one = np.ctypeslib.as_array(base.meshes[0].positions, shape=(n1, 9))
say(f"scene loaded, {n1} triangles per copy")
positions = synth.flattened_copies(one, copies, seed=5, progress=say)
say(f"{positions.shape[0]} triangles, {positions.nbytes / 1e9:.2f} GB of positions")
mesh = N.PtMeshDesc()
mesh.triangle_count = positions.shape[0]
mesh.positions = positions.ctypes.data_as(C.POINTER(C.c_float))
big = N.PtSceneDesc()
for name, _ in N.PtSceneDesc._fields_:
    setattr(big, name, getattr(base, name))
big.meshes = C.pointer(mesh)
big.mesh_count = 1
t = time.time()
gpu = P.HipRenderer.from_desc(C.pointer(big), 0)
say(f"BVH built, widened and uploaded in {time.time() - t:.1f} s")

if check_rays:
    rng = np.random.default_rng(31)
    ext = 80.0
    src = rng.uniform(-ext, ext, (check_rays, 3))
    dst = rng.uniform(-0.7 * ext, 0.7 * ext, (check_rays, 3))
    d = dst - src
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((check_rays, 8), dtype=np.float32)
    rays[:, :3], rays[:, 3:6], rays[:, 7] = src, d, np.finfo(np.float32).max
    t = time.time()
    fast = gpu.trace_closest(rays)
    say(f"BVH search of {check_rays} rays: {time.time() - t:.3f} s")
    t = time.time()
    slow = gpu.trace_closest(rays, brute_force=True)
    say(f"linear scan of {check_rays} rays: {time.time() - t:.3f} s")
    same = fast.tobytes() == slow.tobytes()
    on_mesh = (fast["hit"] == 1) & (fast["tri_index"] != 0xFFFFFFFF)
    say(f"closest hits identical: {same}; on the mesh {on_mesh.mean():.3f}; copies hit "
        f"{np.unique(fast['tri_index'][on_mesh] // n1).size}/{copies}")
    if not same:
        sys.exit(1)

gpu.render(task, P.RenderOpts(seed=0))
say("warm-up render done")
gpu.render(task, P.RenderOpts(seed=0, profile=True))
s = gpu.last_stats
say(f"copies={copies} triangles={positions.shape[0]} {width}x{height} iters={iters} render_ms={s.render_ms:.2f} "
    f"rays={s.rays} searches={s.searches} Mrays/s={s.rays / s.render_ms / 1e3:.1f} "
    f"search_kernel_ms={s.search_kernel_ms:.2f} launches={s.search_kernel_launches} bvh_bytes={s.bvh_bytes}")
gpu.close()
