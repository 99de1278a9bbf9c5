"""Dragon frame time against the number of samples per pixel (what a rank of a frame shared among N GPUs by sample
ranges renders: 64 / N): python3 tools/probe_samples.py"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as P  # noqa: E402
from pathtracing_amd import synth  # noqa: E402

out = os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon")
tp = synth.write_dragon_scene(out, os.path.join(ROOT, "scene_data"))
task = P.RenderTask.from_json(open(tp).read())
r = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
for iters in (2, 4, 8, 16, 32, 64):
    task.config.iterations = iters
    line = f"spp={iters}"
    for name, opts in (("default", P.RenderOpts(seed=0)), ("one_batch", P.RenderOpts(seed=0, single_stream=True))):
        r.render(task, opts)
        ms = 1e9
        for _ in range(4):
            r.render(task, opts)
            ms = min(ms, r.last_stats.render_ms)
        line += f" {name}_ms={ms:.2f} ({r.last_stats.rays / ms / 1e3:.0f} Mrays/s)"
    print(line, flush=True)
