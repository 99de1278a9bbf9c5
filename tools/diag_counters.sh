#!/bin/bash
# tools/diag_counters.sh LIB_BASENAME TAG: issue/stall counters of the trace kernels over one dragon frame with the
# given build of the library (pathtracing_amd/LIB_BASENAME) -> gpurun_out/diag_TAG/ (summed with tools/pmc_summary.py)
set -e
export TMPDIR=/tmp
export PATHTRACING_AMD_LIB=$PWD/pathtracing_amd/$1
export PATHTRACING_AMD_STREAMS=1
OUT=gpurun_out/diag_$2
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES --output-format csv -d $OUT/a -- python3 tools/profile_render.py 64 > $OUT/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_TOTAL_CYCLES --output-format csv -d $OUT/b -- python3 tools/profile_render.py 64 > $OUT/b.log 2>&1
for k in search_kernel walk_bounce_kernel; do echo "[$2 $k]"; python3 tools/pmc_summary.py $OUT/a $k; python3 tools/pmc_summary.py $OUT/b $k; done
