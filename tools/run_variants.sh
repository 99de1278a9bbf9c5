#!/bin/bash
# usage: run_variants.sh lib1 lib2 ...  (names without .so under pathtracing_amd/)
for l in "$@"; do
  PATHTRACING_AMD_LIB=$PWD/pathtracing_amd/$l.so timeout -k 10 150 python3 tools/probe_render.py 64 >> gpurun_out/variants.log 2>&1 || { echo "FAILED $l" >> gpurun_out/variants.log; exit 1; }
done
