#!/bin/bash
# tools/run_config_variants.sh CONFIG LIB...: `bench.py --config CONFIG` (no CPU legs) per build of the library
for l in "${@:2}"; do
  PATHTRACING_AMD_LIB=$PWD/pathtracing_amd/$l.so timeout -k 10 300 python3 bench.py --config $1 --no-cpu-baseline --no-hbm-regime --steps 3 --warmup 1 > gpurun_out/cfg_$l.json 2> gpurun_out/cfg_$l.err || { echo "FAILED $l"; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/cfg_$l.json').read().strip().splitlines()[-1])
print('$l', '$1', round(d['value']), 'Mrays/s', round(d['ms_per_step'],2), 'ms; search launch', round(d['roofline']['launch_ms'],3), 'ms')"
done
