"""Times the benchmark configurations that fit one GPU, one JSON line each:
python3 tools/probe_configs.py [out.json]
  C1  simple_scene.json and scene0.json, 256x256, 4 samples (the plumbing case)
  C2  scene0.json (two textures), 1920x1080, 16 samples
  C3  dragon stand-in, 1920x1080, 64 samples             (what bench.py measures)
  C4  dragon stand-in, 3840x2160, one rank's share of 256 samples over 8 ranks (32), and all 256
  C5  115 instances of the dragon through the scene graph, 3840x2160, 2 samples, whole frame and one rank's rows
Every render is run twice; the second is reported (device time of the render, scene resident)."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as P  # noqa: E402
from pathtracing_amd import distributed as D  # noqa: E402
from pathtracing_amd import synth  # noqa: E402

FIX = os.path.join(ROOT, "scene_data")
rows = []


def timed(name, renderer, task, opts=None, note=""):
    opts = opts or P.RenderOpts(seed=0)
    renderer.render(task, opts)
    renderer.render(task, opts)
    s = renderer.last_stats
    row = {"config": name, "resolution": list(task.camera.resolution), "samples": task.config.iterations,
           "trace_depth": task.config.trace_depth, "rays": int(s.rays), "searches": int(s.searches),
           "render_ms": round(s.render_ms, 3), "mrays_per_s": round(s.rays / s.render_ms / 1e3, 1), "note": note}
    rows.append(row)
    print(json.dumps(row), flush=True)


def fixture_task(scene, w, h, spp):
    t = P.RenderTask.from_json(open(os.path.join(FIX, "render_task.json")).read())
    t.scene = scene
    t.camera.resolution = [w, h]
    t.config = P.Config(trace_depth=8, iterations=spp)
    return t


for scene in ("./simple_scene.json", "./scene0.json"):
    r = P.HipRenderer.init(P.Scene.load(FIX, scene), 0)
    timed(f"C1 {scene[2:]}", r, fixture_task(scene, 256, 256, 4), note="65 k pixels: launch-bound")
    if scene == "./scene0.json":
        timed("C2 scene0.json", r, fixture_task(scene, 1920, 1080, 16))
    r.close()

out = os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon")
tp = synth.write_dragon_scene(out, FIX)
task = P.RenderTask.from_json(open(tp).read())
dragon = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
timed("C3 dragon", dragon, task)
task.camera.resolution = [3840, 2160]
task.config.iterations = 256
sub, opts = D.shard_opts(task, 8, 3, "samples", seed=0)
timed("C4 dragon 4K, rank 3 of 8 by sample range", dragon, sub, opts, note="32 of the 256 samples")
timed("C4 dragon 4K, all 256 samples on one GPU", dragon, task)
dragon.close()

with open(os.path.join(out, "instanced.json"), "w") as f:
    json.dump(synth.instanced_dragon_scene_json(115, seed=3), f)
task = P.RenderTask.from_json(open(tp).read())
task.scene = "./instanced.json"
task.camera.position = [0.0, 10.0, 110.0]
task.camera.focal_length = 110.0
task.camera.resolution = [3840, 2160]
task.config.iterations = 2
inst = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
timed("C5 115 instances (1.0e8 triangles instanced), 4K", inst, task)
sub, opts = D.shard_opts(task, 8, 3, "rows", seed=0)
timed("C5 115 instances, 4K, rank 3 of 8 by rows", inst, sub, opts)
inst.close()

if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(rows, f, indent=1)
