"""bench.py's own launch of its ranks: `python bench.py --gpus N` from a plain shell must
start N fresh processes, shard, reduce onto rank 0 and print one JSON line.  On CPU the
render is a stand-in (--stub-render: every sample adds 1 to the pixels it covers), so a
resolved frame mean of exactly 1 says that every (pixel, sample) was rendered once and
the buffers were summed and scaled correctly.  On the GPU box the same launch runs the
real render on two ranks sharing GPU 0 (--rehearse: gloo, host-staged sum)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers as H

BENCH = os.path.join(H.ROOT, "bench.py")


def _run(*flags, timeout=600):
    env = dict(os.environ)
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(var, None)
    proc = subprocess.run([sys.executable, BENCH, *flags], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.parametrize("config,scaling", [("c3", "weak"), ("c4", "strong"), ("c5", "strong")])
def test_plain_invocation_starts_its_ranks(config, scaling):
    line = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--stub-render", "--config", config)
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["scaling"] == scaling and line["config"]["ranks"] == 2 and len(line["config"]["devices"]) == 2
    assert line["config"]["frame_mean"] == 1.0
    per_step = 64 * 40 * (4 if config == "c3" else 8)  # c3 stand-in: 2 samples per rank; c4, c5: 8 in all
    assert line["config"]["rays_per_step"] == per_step and line["value"] > 0


def test_single_rank_stub_line_has_the_contract_fields():
    line = _run("--stub-render", "--steps", "1", "--warmup", "0")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line
    assert line["n_gpus"] == 1 and line["config"]["frame_mean"] == 1.0 and line["vs_baseline"] is None


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_render_the_frame():
    """The real chain on device buffers under two ranks: pt_render_accumulate_device on each
    rank's sample range, reduce onto rank 0, pt_resolve_device; against one process rendering
    all samples (sums in another order: 1e-5 relative)."""
    import pathtracing_amd as P
    from pathtracing_amd import synth

    line = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--rehearse", "--no-cpu-baseline", "--no-hbm-regime")
    assert line["n_gpus"] == 2 and "REHEARSAL" in line["config"]["sharding"]
    assert line["roofline"] is not None and 0.0 < line["roofline"]["frac"]
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "pathtracing_amd_dragon")
    task = H.load_task(synth.write_dragon_scene(out, H.SCENE_DATA))
    task.camera.resolution = [1920, 1080]
    task.config = P.Config(trace_depth=8, iterations=128)
    gpu = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
    img = gpu.render(task, P.RenderOpts(seed=0))
    assert gpu.last_stats.rays == line["config"]["rays_per_step"]
    mean = float(np.nanmean(img))
    assert abs(mean - line["config"]["frame_mean"]) <= 1e-5 * mean
    gpu.close()
