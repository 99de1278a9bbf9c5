#!/bin/bash
# Run on the GPU box from the repository root: tools/compare_search_counters.sh LIB_A LIB_B
# SQ counters of the BVH search kernel over one dragon frame for two builds of the library.
set -e
export TMPDIR=/tmp
COUNTERS="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES"
for lib in "$@"; do
  name=$(basename $lib .so)
  export PATHTRACING_AMD_LIB=$PWD/$lib
  OUT=gpurun_out/cmp_$name
  rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 120 rocprofv3 --pmc $COUNTERS --output-format csv -d $OUT/sq -- python3 tools/profile_render.py 64 > $OUT/sq.log 2>&1
  echo "[$name]"
  python3 tools/pmc_summary.py $OUT/sq search_
done
