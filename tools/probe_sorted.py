"""How much ray order matters to the search kernel: third-segment rays of the dragon scene traced through the
wavefront route in several orders.  Run under rocprofv3 --kernel-trace; tools/probe_sorted_report.py lists the
search_kernel durations in launch order (two launches per order, labels printed here)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as P  # noqa: E402
from pathtracing_amd import synth  # noqa: E402
from tests import helpers as H  # noqa: E402

out = os.path.join(tempfile.gettempdir(), "pathtracing_amd_dragon")
tp = synth.write_dragon_scene(out, os.path.join(ROOT, "scene_data"))
task = P.RenderTask.from_json(open(tp).read())
r = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
ys, xs = np.mgrid[0:1080, 0:1920]
px = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.uint32)
px = np.concatenate([px, px])
rays = P.generate_rays(task.camera, 0, px, np.arange(px.shape[0], dtype=np.uint64) // (1920 * 1080))
for seg in range(2):
    hits = r.trace_closest(rays)
    rays = H.secondary_rays(hits, rays, seed=seg)
n = rays.shape[0]
print("third-segment rays:", n, flush=True)


def morton(o, bits):
    lo, hi = o.min(0), o.max(0)
    q = np.clip(((o - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    key = np.zeros(o.shape[0], dtype=np.int64)
    for b in range(bits):
        for a in range(3):
            key |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return key


octant = (rays[:, 3] < 0).astype(np.int64) | ((rays[:, 4] < 0).astype(np.int64) << 1) | ((rays[:, 5] < 0).astype(np.int64) << 2)
rng = np.random.default_rng(1)
orders = {
    "natural": np.arange(n),
    "shuffled": rng.permutation(n),
    "octant": np.argsort(octant, kind="stable"),
    "morton5": np.argsort(morton(rays[:, 0:3], 5), kind="stable"),
    "octant+morton5": np.argsort((octant << 15) | morton(rays[:, 0:3], 5), kind="stable"),
    "morton5+octant": np.argsort((morton(rays[:, 0:3], 5) << 3) | octant, kind="stable"),
    "morton8+octant": np.argsort((morton(rays[:, 0:3], 8) << 3) | octant, kind="stable"),
}
base = r.trace_closest(rays, route="wavefront")
for name, order in orders.items():
    sel = np.ascontiguousarray(rays[order])
    for _ in range(2):
        got = r.trace_closest(sel, route="wavefront")
    assert np.array_equal(got["t"].view(np.uint32), base["t"][order].view(np.uint32))
    print("order", name, flush=True)
