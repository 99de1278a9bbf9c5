#!/bin/bash
# tools/diag_search.sh LIB...: one rocprofv3 --pmc pass per build (pathtracing_amd/LIB.so) over one dragon frame, issue
# counters of search_kernel summed over its launches -> gpurun_out/diag_search.txt
export TMPDIR=/tmp
export PATHTRACING_AMD_STREAMS=1
for l in "$@"; do
  OUT=gpurun_out/diags_$l
  mkdir -p $OUT
  PATHTRACING_AMD_LIB=$PWD/pathtracing_amd/$l.so timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD --output-format csv -d $OUT -- python3 tools/profile_render.py 64 > $OUT.log 2>&1 || exit 1
  echo "[$l search_kernel]" >> gpurun_out/diag_search.txt
  python3 tools/pmc_summary.py $OUT search_kernel >> gpurun_out/diag_search.txt
done
