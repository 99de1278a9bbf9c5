"""Render rate of the dragon workload against the batch size, with one to four batches in flight
(each in a workspace of its own, on the library's side streams): python3 tools/probe_batch.py [iterations]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as P  # noqa: E402
from pathtracing_amd import synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out = os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon")
tp = synth.write_dragon_scene(out, os.path.join(ROOT, "scene_data"))
task = P.RenderTask.from_json(open(tp).read())
task.config.iterations = iters
r = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
for mi in (4, 8, 16, 32, 64, 128):
    for flight in (1, 2, 3, 4):
        os.environ["PATHTRACING_AMD_STREAMS"] = str(flight)
        o = P.RenderOpts(seed=0, paths_per_batch=mi << 20)
        r.render(task, o)
        best = 1e9
        for _ in range(3):
            r.render(task, o)
            best = min(best, r.last_stats.render_ms)
        s = r.last_stats
        print(f"paths_per_batch={mi}Mi in flight {flight} render_ms={best:.2f} "
              f"Mrays/s={s.rays / best / 1e3:.1f}", flush=True)
