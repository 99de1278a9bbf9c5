"""Wave lifetimes of one search_kernel launch (library built with -DPT_WAVE_TIMES): dragon frame cut after DEPTH segments."""
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
for depth, spp in ((1, 64), (1, 32), (2, 32)):
    task.config = P.Config(trace_depth=depth, iterations=spp)
    o = P.RenderOpts(seed=0, paths_per_batch=128 << 20, single_stream=True, count_visits=True)
    r.render(task, o)
    r.render(task, o)
    s = r.last_stats
    t0 = (1 << 62) - s.node_visits
    span = s.triangle_tests - t0
    waves = s.wave_leaf_steps
    print(f"depth={depth} spp={spp} launches~{depth} waves={waves} span_ticks={span} mean_lifetime_ticks={s.wave_node_steps / max(waves, 1):.0f} "
          f"(wall clock ticks of 10 ns) mean/span={s.wave_node_steps / max(waves, 1) / max(span, 1):.3f} jobs={s.wave_leaf_rounds} searches={s.searches}", flush=True)
