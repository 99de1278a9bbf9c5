#!/bin/bash
# tools/diag_ifetch.sh LIB...: instruction fetches, branches and issue counters of search_kernel per build
export TMPDIR=/tmp
export PATHTRACING_AMD_STREAMS=1
for l in "$@"; do
  OUT=gpurun_out/diagi_$l
  mkdir -p $OUT
  PATHTRACING_AMD_LIB=$PWD/pathtracing_amd/$l.so timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -- python3 tools/profile_render.py 64 > $OUT.log 2>&1 || exit 1
  echo "[$l search_kernel]"; python3 tools/pmc_summary.py $OUT search_kernel
done
