#!/bin/bash
# tools/diag_frontend.sh: instruction-fetch, branch, scalar-cache and TA-queue counters of the trace kernels over one dragon
# frame (two rocprofv3 --pmc passes) -> gpurun_out/diag_frontend.txt
export TMPDIR=/tmp
export PATHTRACING_AMD_STREAMS=1
OUT=gpurun_out/diagf
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SMEM --output-format csv -d $OUT/a -- python3 tools/profile_render.py 64 > $OUT/a.log 2>&1 || { tail -3 $OUT/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES --output-format csv -d $OUT/b -- python3 tools/profile_render.py 64 > $OUT/b.log 2>&1 || { tail -3 $OUT/b.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES --output-format csv -d $OUT/c -- python3 tools/profile_render.py 64 > $OUT/c.log 2>&1 || { tail -3 $OUT/c.log; exit 1; }
for k in search_kernel walk_bounce_kernel; do echo "[$k]"; for p in a b c; do python3 tools/pmc_summary.py $OUT/$p $k; done; done > gpurun_out/diag_frontend.txt
cat gpurun_out/diag_frontend.txt
