#!/bin/bash
# tools/run_hbm_variants.sh LIB...: the HBM-regime record of bench.py (flattened 2.1e7-triangle scene) per build of the library
for l in "$@"; do
  PATHTRACING_AMD_LIB=$PWD/pathtracing_amd/$l.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/hbm_$l.json 2> gpurun_out/hbm_$l.err || { echo "FAILED $l"; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/hbm_$l.json').read().strip().splitlines()[-1])
h=d['roofline_hbm_regime']
print('$l', 'frame', round(d['ms_per_step'],2), 'ms | hbm regime: launch', round(h['launch_ms'],3), 'ms', round(h['mrays_per_s']), 'Mrays/s nodes/search', round(h['node_visits_per_search'],2), 'lanes/node step', round(h['lanes_per_node_step'],1))"
done
