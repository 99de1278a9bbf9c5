import os, sys, tempfile, ctypes
ROOT = "/root/repo" if os.path.isdir("/root/repo/pathtracing_amd") else os.getcwd()
sys.path.insert(0, os.getcwd())
import pathtracing_amd as P
from pathtracing_amd import synth
out = os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon")
tp = synth.write_dragon_scene(out, "scene_data")
task = P.RenderTask.from_json(open(tp).read())
r = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
lib = P._native.lib
buf = (ctypes.c_ulonglong * 8)()
for depth in (1, 2, 3, 4, 8):
    task.config = P.Config(trace_depth=depth, iterations=8)
    lib.pt_debug_step_counts(buf, 1)
    r.render(task, P.RenderOpts(seed=0))
    lib.pt_debug_step_counts(buf, 0)
    il, ip, ll, lp, jobs, mx = [buf[i] for i in range(6)]
    print(f"depth {depth}: render_ms {r.last_stats.render_ms:.1f} searches {r.last_stats.searches} inner lane-steps {il} ({il/max(1,r.last_stats.searches):.1f}/search) passes {ip} (lanes/pass {il/max(1,ip):.1f}) leaf lane-steps {ll} ({ll/max(1,r.last_stats.searches):.2f}/search) passes {lp} max inner passes per wave {mx}")
