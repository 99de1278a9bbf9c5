"""Dragon frame timing for kernel experiments: python3 tools/probe_render.py [iterations]
Prints the best of three renders (two batches in flight, as a default render), the same with one
batch at a time, and the summed search_kernel time of a profiled single-stream render."""
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


def best(opts, n=3):
    r.render(task, opts)
    ms = 1e9
    for _ in range(n):
        r.render(task, opts)
        ms = min(ms, r.last_stats.render_ms)
    return ms, r.last_stats


ms2, s = best(P.RenderOpts(seed=0))
ms1, _ = best(P.RenderOpts(seed=0, paths_per_batch=128 << 20, single_stream=True))
_, sp = best(P.RenderOpts(seed=0, paths_per_batch=128 << 20, profile=True), n=1)
_, sc = best(P.RenderOpts(seed=0, paths_per_batch=128 << 20, count_visits=True), n=1)
print(f"lib={os.path.basename(P._native.LIB_PATH)} iters={iters} rays={s.rays} default_ms={ms2:.2f} ({s.rays / ms2 / 1e3:.0f} Mrays/s) "
      f"one_batch_ms={ms1:.2f} search_kernel_ms={sp.search_kernel_ms:.2f} other_ms={sp.render_ms - sp.search_kernel_ms:.2f} "
      f"nodes/search={sc.node_visits / sc.searches:.2f} tris/search={sc.triangle_tests / sc.searches:.2f} "
      f"lanes/node_step={sc.node_visits / max(sc.wave_node_steps, 1):.1f} lanes/tri_round={sc.triangle_tests / max(sc.wave_leaf_rounds, 1):.1f} "
      f"node_steps={sc.wave_node_steps} leaf_steps={sc.wave_leaf_steps} tri_rounds={sc.wave_leaf_rounds}", flush=True)
