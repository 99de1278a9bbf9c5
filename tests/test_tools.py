"""The measuring tools that feed DESIGN.md's figures keep working: the host-side tree-quality probe builds against
the library's builder and restates the device's search, and the issue-mix script finds the kernels in the
library's device code (no GPU needed for either)."""
import json
import os
import shutil
import subprocess
import sys

import pytest

from pathtracing_amd import synth
from tests import helpers as H


@pytest.mark.skipif(shutil.which("g++") is None, reason="no C++ compiler")
def test_tree_quality_probe_builds_and_counts(tmp_path):
    exe = str(tmp_path / "bvh_quality")
    csrc = os.path.join(H.ROOT, "pathtracing_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", csrc, "-I", os.path.join(H.ROOT, "include"),
                           os.path.join(H.ROOT, "tools", "bvh_quality.cpp"), os.path.join(csrc, "host", "bvh.cpp"),
                           os.path.join(csrc, "host", "obj_loader.cpp"), "-pthread", "-o", exe])
    verts, tris = synth.knot_mesh(120, 48)
    obj = str(tmp_path / "m.obj")
    synth.write_obj(obj, verts, tris)
    out = subprocess.check_output([exe, obj, "2000"], text=True)
    assert "triangles 11520" in out and "4-wide nodes with exact boxes" in out and "8-wide nodes with exact boxes" in out
    line = [ln for ln in out.splitlines() if ln.startswith("all ")][0]
    nodes = float(line.split("searches:")[1].split("node records")[0])
    assert 3.0 < nodes < 40.0
    # every tuning gives a searchable tree: the counts move, the probe still runs
    env = dict(os.environ, PATHTRACING_AMD_BVH_TRI_COST="2.0")
    other = subprocess.check_output([exe, obj, "2000"], text=True, env=env)
    assert other != out


def test_issue_mix_finds_the_kernels():
    lib = os.path.join(H.ROOT, "pathtracing_amd", "libpathtracing_amd.so")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    out = subprocess.check_output([sys.executable, os.path.join(H.ROOT, "tools", "issue_mix.py"), lib, "--json"], text=True)
    rec = json.loads(out)
    search = [v for k, v in rec.items() if "search_kernel" in k]
    assert len(search) == 1 and search[0]["valu"] > 200 and 0.2 < search[0]["slow_class_share"] < 0.9
    assert search[0]["node_step_block"]["valu"] > 60
    assert any("walk_bounce_kernel" in k for k in rec)
