import sys, time, json, os
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracing_amd as P
from pathtracing_amd import synth
out = '/tmp/dragon_probe'
t0 = time.time(); tp = synth.write_dragon_scene(out, 'scene_data'); print('gen', time.time() - t0, flush=True)
task = P.RenderTask.from_json(open(tp).read())
t0 = time.time(); sc = P.Scene.load(out, task.scene); print('load', time.time() - t0, flush=True)
t0 = time.time(); r = P.HipRenderer.init(sc, 0); print('init(bvh+upload)', time.time() - t0, flush=True)
for it in (1, 8, 64):
    task.config.iterations = it
    for rep in range(2):
        t0 = time.time(); img = r.render(task, P.RenderOpts(seed=0, profile=(rep == 1))); dt = time.time() - t0
        s = r.last_stats
        print(f'iters={it} wall={dt:.3f}s render_ms={s.render_ms:.1f} rays={s.rays} Mrays/s={s.rays / s.render_ms / 1e3:.1f} '
              f'search_ms={s.search_kernel_ms:.1f} launches={s.search_kernel_launches} searches={s.searches} mean={img.mean():.4f}', flush=True)
