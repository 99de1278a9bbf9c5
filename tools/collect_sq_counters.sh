#!/bin/bash
# Run on the GPU box from the repository root: tools/collect_sq_counters.sh ROUND
# SQ counters of the four trace kernels over one dragon frame (tools/profile_render.py 64 renders it twice)
# -> profiles/rR_sq_counters.txt
set -e
R=${1:-1}
export TMPDIR=/tmp
OUT=gpurun_out/sq_r$R
mkdir -p $OUT profiles
COUNTERS="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --pmc $COUNTERS --output-format csv -d $OUT/sq -- python3 tools/profile_render.py 64 > $OUT/sq.log 2>&1
{
  echo "# rocprofv3 --pmc $COUNTERS -- python3 tools/profile_render.py 64"
  echo "# dragon stand-in 1920x1080, 64 spp, depth 8 (one batch per render: 8 launches of each trace kernel, two renders); sums over launches"
  for k in search_kernel walk_bounce_kernel camera_kernel; do echo "[$k]"; python3 tools/pmc_summary.py $OUT/sq $k; done
} > profiles/r${R}_sq_counters.txt
cat profiles/r${R}_sq_counters.txt
