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


@pytest.mark.parametrize("config", ["c3", "c4", "c5"])
def test_plain_invocation_starts_its_ranks(config):
    line = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--stub-render", "--config", config)
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    # every configuration renders a fixed frame, cut over the ranks: the frame the metric names, on more GPUs
    assert line["scaling"] == "strong" and line["config"]["ranks"] == 2 and len(line["config"]["devices"]) == 2
    assert line["config"]["frame_mean"] == 1.0
    assert line["config"]["rays_per_step"] == 64 * 40 * 8 and line["value"] > 0  # the stand-in frame: 8 samples in all
    assert line["config"]["reduce_ms"] >= 0.0
    if config == "c3":  # ... and beside it the same loop with that many samples on every rank
        weak = line["extra"]["weak_64spp_per_gpu"]
        assert weak["scaling"] == "weak" and weak["samples_per_pixel_in_all"] == 16
        assert weak["rays_per_step"] == 64 * 40 * 16 and weak["frame_mean"] == 1.0 and weak["value"] > 0
    else:
        assert "extra" not in line


def test_result_line_is_found_among_foreign_output():
    """The ranks share one pipe with the collective library's banners: whatever else is on it --
    valid JSON that is no object, half a line -- the parser neither raises nor loses rank 0's line."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_module", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    result = json.dumps({"metric": "Mrays/s dragon 1080p 64spp", "value": 1.0})
    noise = ["1", "null", "[1]", '"x"', "true", "3.5", '{"metric": "Mr', "[Gloo] Rank 0 is connected to 1 peer ranks. Expected number of connected peer ranks is : ",
             '{"other": 1}', "", "   "]
    line, foreign = bench.pick_result_line("\n".join(noise[:5] + [result] + noise[5:]))
    assert line == result and len(foreign) == len([n for n in noise if n.strip()])
    assert bench.pick_result_line("\n".join(noise)) == (None, [n for n in noise if n.strip()])
    assert bench.pick_result_line("") == (None, [])
    # the last result line wins (a rank that printed twice)
    second = json.dumps({"metric": "m", "value": 2.0})
    assert bench.pick_result_line(result + "\n1\n" + second)[0] == second


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
    gpu = P.HipRenderer.init(P.Scene.load(out, task.scene), 0)
    for iters, part in ((64, line["config"]), (128, line["extra"]["weak_64spp_per_gpu"])):
        task.config = P.Config(trace_depth=8, iterations=iters)
        img = gpu.render(task, P.RenderOpts(seed=0))
        assert gpu.last_stats.rays == part["rays_per_step"]
        mean = float(np.nanmean(img))
        assert abs(mean - part["frame_mean"]) <= 1e-5 * mean
    gpu.close()


@pytest.mark.gpu
def test_one_process_drives_two_scene_handles():
    """--inproc: the same steps through pt_render_multi (threads, streams, cached buffers, the peer-copy sum:
    both handles live on GPU 0); rays and frame mean against the two-process rehearsal's, which renders the
    same frame through torch.distributed."""
    common = ("--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse", "--no-cpu-baseline", "--no-hbm-regime")
    one = _run(*common, "--inproc")
    assert one["n_gpus"] == 2 and one["config"]["processes"] == 1 and "pt_render_multi" in one["config"]["sharding"]
    assert one["scaling"] == "strong" and one["config"]["reduce_ms"] > 0.0 and one["roofline"]["frac"] > 0.0
    two = _run(*common)
    assert two["config"]["processes"] == 2
    assert one["config"]["rays_per_step"] == two["config"]["rays_per_step"]
    assert abs(one["config"]["frame_mean"] - two["config"]["frame_mean"]) <= 1e-5 * two["config"]["frame_mean"]
    for line in (one, two):
        weak = line["extra"]["weak_64spp_per_gpu"]
        assert weak["samples_per_pixel_in_all"] == 128 and weak["rays_per_step"] > 1.9 * line["config"]["rays_per_step"]
