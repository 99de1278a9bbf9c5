#!/bin/bash
# Run on the GPU box from the repository root: tools/collect_profiles.sh ROUND
# Writes the rocprofv3 summaries behind bench.py's roofline object under profiles/:
#   rR_bench_kernel_stats.csv   per-kernel time of `python3 bench.py --no-cpu-baseline`
#   rR_pmc_fetch.csv / rR_pmc_write.csv   FETCH_SIZE and WRITE_SIZE per launch (separate passes)
#   hbm_traffic.json            HBM-side bytes per search_kernel launch
set -e
R=${1:-1}
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r$R
mkdir -p $OUT profiles
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/trace.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv profiles/r${R}_bench_kernel_stats.csv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write.log 2>&1
python3 tools/summarize_traffic.py $OUT $R
