import os, sys, tempfile, ctypes
sys.path.insert(0, os.getcwd())
import pathtracing_amd as P
from pathtracing_amd import synth
out = os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon")
tp = synth.write_dragon_scene(out, "scene_data")
task = P.RenderTask.from_json(open(tp).read())
r = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
lib = P._native.lib
buf = (ctypes.c_ulonglong * 8)()
for depth in (1, 8):
    task.config = P.Config(trace_depth=depth, iterations=8)
    lib.pt_debug_step_counts(buf, 1)
    r.render(task, P.RenderOpts(seed=0))
    lib.pt_debug_step_counts(buf, 0)
    c = [buf[i] for i in range(6)]
    print(f"depth {depth}: inner passes {c[1]} lanes at inner {c[0]/c[1]:.1f} at leaf {c[2]/c[1]:.1f} idle/done {c[3]/c[1]:.1f} | leaf passes {c[5]} lanes {c[4]/max(1,c[5]):.1f}")
