"""python3 tools/probe_depth.py DEPTH: device time of a dragon frame cut off after DEPTH path segments (kernel experiments)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as P  # noqa: E402
from pathtracing_amd import synth  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
out = os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon")
tp = synth.write_dragon_scene(out, os.path.join(ROOT, "scene_data"))
task = P.RenderTask.from_json(open(tp).read())
task.config = P.Config(trace_depth=depth, iterations=64)
r = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
o = P.RenderOpts(seed=0, paths_per_batch=128 << 20, single_stream=True)
r.render(task, o)
best = 1e9
for _ in range(3):
    r.render(task, o)
    best = min(best, r.last_stats.render_ms)
print(f"lib={os.path.basename(P._native.LIB_PATH)} depth={depth} render_ms={best:.2f} rays={r.last_stats.rays}", flush=True)
