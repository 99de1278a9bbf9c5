#!/bin/bash
# Run on the GPU box from the repository root: tools/collect_profiles.sh ROUND   (raw rocprofv3 output under gpurun_out/)
# then, where the repository lives:              tools/collect_profiles.sh ROUND summarize
# The second step writes the summaries behind bench.py's roofline objects under profiles/:
#   rR_bench_kernel_stats.csv              per-kernel time of `python3 bench.py --no-cpu-baseline --no-hbm-regime`
#   rR_bench_kernel_stats_one_batch.csv    the same with PATHTRACING_AMD_STREAMS=1: no two kernels overlap, so the average
#                                          duration of search_kernel is the launch time bench.py measures with events
#   rR_pmc_fetch.csv / rR_pmc_write.csv    FETCH_SIZE and WRITE_SIZE per launch (separate passes), dragon frame
#   hbm_traffic.json                       HBM-side bytes per search_kernel launch, dragon frame
#   rR_sq_counters.txt, sq_counters.json   SQ counters of the trace kernels, dragon frame
#   rR_search_issue_mix.txt                the kernels' vector instructions by issue class (tools/issue_mix.py)
# Every .json carries kernel_fingerprint: the device code the counters were taken with.
#   rR_pmc_*_flattened.csv, hbm_traffic_flattened.json, rR_flattened_kernel_stats.csv   the 2.1e7-triangle flattened scene
set -e
R=${1:-2}
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r$R
FL=gpurun_out/flattened_r$R
if [ "$2" = "summarize" ]; then
  mkdir -p profiles
  cp $OUT/trace/*/*kernel_stats.csv profiles/r${R}_bench_kernel_stats.csv
  grep '^{"metric"' $OUT/trace.log | tail -1 > profiles/r${R}_bench_line_under_rocprof.json
  cp $OUT/trace_one/*/*kernel_stats.csv profiles/r${R}_bench_kernel_stats_one_batch.csv
  python3 tools/summarize_traffic.py $OUT $R
  python3 tools/summarize_sq.py $OUT/sq $R > /dev/null
  cp $FL/trace/*/*kernel_stats.csv profiles/r${R}_flattened_kernel_stats.csv
  python3 tools/summarize_traffic.py $FL $R flattened
  cp $OUT/issue_mix.txt profiles/r${R}_search_issue_mix.txt
  exit 0
fi
mkdir -p $OUT $FL
# which kernels the counters below belong to: bench.py replays them only into runs of the same device code
python3 -c "from pathtracing_amd import _native as N; print(N.kernel_fingerprint())" | tee $OUT/kernel_fingerprint.txt > $FL/kernel_fingerprint.txt
python3 tools/issue_mix.py --json > $OUT/issue_mix.json
python3 tools/issue_mix.py > $OUT/issue_mix.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-hbm-regime > $OUT/trace.log 2>&1
export PATHTRACING_AMD_STREAMS=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_one -- python3 bench.py --no-cpu-baseline --no-hbm-regime > $OUT/trace_one.log 2>&1
unset PATHTRACING_AMD_STREAMS
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/profile_render.py 64 > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/profile_render.py 64 > $OUT/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 tools/profile_render.py 64 > $OUT/sq.log 2>&1
CMD="python3 -u tools/probe_flattened.py 24 3840 2160 2 0"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $FL/trace -- $CMD > $FL/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $FL/fetch -- $CMD > $FL/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $FL/write -- $CMD > $FL/write.log 2>&1
