#!/bin/bash
# Run on the GPU box from the repository root: tools/collect_flattened_profiles.sh ROUND
# rocprofv3 passes over the flattened-instances workload (115 copies = 1.0e8 triangles,
# 3840x2160, 2 samples, depth 8; tools/probe_flattened.py renders it twice):
#   rR_flattened_kernel_stats.csv   per-kernel time
#   rR_flattened_counters.txt       FETCH_SIZE, WRITE_SIZE and SQ counters of search_kernel, summed over launches
set -e
R=${1:-1}
export TMPDIR=/tmp
OUT=gpurun_out/flattened_r$R
CMD="python3 -u tools/probe_flattened.py 115 3840 2160 2 0"
mkdir -p $OUT profiles
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv profiles/r${R}_flattened_kernel_stats.csv
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
{
  echo "# rocprofv3 --pmc <one pass per group> -- $CMD"
  echo "# 1.0e8 world-space triangles in one mesh, 3840x2160, 2 samples, depth 8; two renders = 16 search_kernel launches; sums over launches (FETCH_SIZE / WRITE_SIZE in KiB, raw)"
  tail -1 $OUT/trace.log
  for p in fetch write sq; do python3 tools/pmc_summary.py $OUT/$p search_kernel; done
} > profiles/r${R}_flattened_counters.txt
cat profiles/r${R}_flattened_counters.txt
