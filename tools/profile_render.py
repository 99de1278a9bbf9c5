"""Short dragon render for counter profiling: python3 tools/profile_render.py [iterations] [width height]."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as P  # noqa: E402
from pathtracing_amd import synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8
out = os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon")
tp = synth.write_dragon_scene(out, os.path.join(ROOT, "scene_data"))
task = P.RenderTask.from_json(open(tp).read())
if len(sys.argv) > 3:
    task.camera.resolution = [int(sys.argv[2]), int(sys.argv[3])]
task.config.iterations = iters
r = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
img = r.render(task, P.RenderOpts(seed=0))
s = r.last_stats
print(f"iters={iters} render_ms={s.render_ms:.2f} rays={s.rays} Mrays/s={s.rays / s.render_ms / 1e3:.1f}", flush=True)
