#!/bin/bash
# tools/run_tree_variants.sh [--hbm] "NAME=VALUE ..." ...: bench.py once per setting of the tree builder
# (PATHTRACING_AMD_BVH_<NAME>, host/bvh.cpp: default_bvh_tuning; a NAME that starts with PATHTRACING_AMD_ is taken as
# it stands, for the library's other knobs), one line of its search-kernel record each.
# With --hbm the flattened 2.1e7-triangle scene's record is kept beside the frame's.
mkdir -p gpurun_out
hbm=--no-hbm-regime
if [ "$1" = "--hbm" ]; then hbm=; shift; fi
i=0
for cfg in "$@"; do
  i=$((i+1))
  envs=()
  for kv in $cfg; do case $kv in PATHTRACING_AMD_*) envs+=("$kv");; *) envs+=("PATHTRACING_AMD_BVH_$kv");; esac; done
  env "${envs[@]}" timeout -k 10 400 python3 bench.py --no-cpu-baseline $hbm --steps 3 --warmup 1 > gpurun_out/tree_$i.json 2> gpurun_out/tree_$i.err || { echo "FAILED $cfg"; tail -5 gpurun_out/tree_$i.err; exit 1; }
  python3 - "$cfg" gpurun_out/tree_$i.json <<'PY' | tee -a gpurun_out/tree_variants.txt
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
line = (f"{sys.argv[1]:44s} frame {d['ms_per_step']:7.2f} ms {d['value']:6.0f} Mrays/s | search launch {r['launch_ms']:.3f} ms "
        f"share {r['search_kernel_share_of_render']:.3f} nodes {r['node_visits_per_search']:.2f} tris {r['triangle_tests_per_search']:.2f} "
        f"lanes/node step {r['lanes_per_node_step']:.1f} lanes/tri round {r['lanes_per_triangle_round']:.1f} build {d['config']['bvh_build_s']:.2f} s")
h = d.get("roofline_hbm_regime")
if h:
    line += (f" || hbm regime: launch {h['launch_ms']:.3f} ms {h['mrays_per_s']:.0f} Mrays/s nodes {h['node_visits_per_search']:.2f} "
             f"tris {h['triangle_tests_per_search']:.2f} frac {h['frac']:.3f} setup {h['scene_setup_s']:.1f} s")
print(line)
PY
done
