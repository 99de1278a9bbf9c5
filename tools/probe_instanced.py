"""Times a render of N instanced dragons (scene-graph instances of one mesh): python3 tools/probe_instanced.py [count] [iterations]."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as P  # noqa: E402
from pathtracing_amd import synth  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 9
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
out = os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon")
tp = synth.write_dragon_scene(out, os.path.join(ROOT, "scene_data"))
with open(os.path.join(out, "instanced.json"), "w") as f:
    json.dump(synth.instanced_dragon_scene_json(count, seed=3), f)
task = P.RenderTask.from_json(open(tp).read())
task.scene = "./instanced.json"
task.camera.position = [0.0, 10.0, 110.0]
task.camera.focal_length = 110.0
task.config.iterations = iters
r = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
r.render(task, P.RenderOpts(seed=0))
img = r.render(task, P.RenderOpts(seed=0))
s = r.last_stats
print(f"instances={count} iters={iters} render_ms={s.render_ms:.2f} rays={s.rays} searches={s.searches} Mrays/s={s.rays / s.render_ms / 1e3:.1f}", flush=True)
