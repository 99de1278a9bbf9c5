"""Dragon frame time against the number of batches in flight and the batch size:
python3 tools/probe_flight.py [iterations]   (one process per setting: the stream count is read from the environment
when the library first renders)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, tempfile
sys.path.insert(0, %r)
import pathtracing_amd as P
from pathtracing_amd import synth
iters, budget = int(sys.argv[1]), int(sys.argv[2])
out = os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon")
tp = synth.write_dragon_scene(out, os.path.join(%r, "scene_data"))
task = P.RenderTask.from_json(open(tp).read())
task.config.iterations = iters
r = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
opts = P.RenderOpts(seed=0, paths_per_batch=budget << 20)
r.render(task, opts)
ms = 1e9
for _ in range(4):
    r.render(task, opts)
    ms = min(ms, r.last_stats.render_ms)
print(f"streams={os.environ.get('PATHTRACING_AMD_STREAMS')} batch={budget}M frame_ms={ms:.2f}", flush=True)
""" % (ROOT, ROOT)

iters = sys.argv[1] if len(sys.argv) > 1 else "64"
for streams in ("1", "2", "3", "4"):
    for budget in ("64", "32", "16", "8"):
        env = dict(os.environ, PATHTRACING_AMD_STREAMS=streams)
        subprocess.run([sys.executable, "-c", CHILD, iters, budget], env=env, check=True, timeout=200)
