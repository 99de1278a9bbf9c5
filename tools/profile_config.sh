#!/bin/bash
# tools/profile_config.sh CONFIG: per-kernel time of `python3 bench.py --config CONFIG` (no CPU legs), one batch at a time
export TMPDIR=/tmp
export PATHTRACING_AMD_STREAMS=1
OUT=gpurun_out/profile_$1
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --config $1 --no-cpu-baseline --no-hbm-regime > $OUT/trace.log 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.3:
            print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>4s} avg_ms {float(r["AverageNs"]) / 1e6:8.3f} share {r["Percentage"]}')
PY
