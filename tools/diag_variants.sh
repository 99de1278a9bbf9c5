#!/bin/bash
# VALU instructions of walk_bounce_kernel's first launch (depth-2 frame: identical input for every variant) per build
export TMPDIR=/tmp
export PATHTRACING_AMD_STREAMS=1
for l in "$@"; do
  OUT=gpurun_out/diagv_$l
  mkdir -p $OUT
  PATHTRACING_AMD_LIB=$PWD/pathtracing_amd/$l.so timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT -- python3 tools/probe_depth.py 2 > $OUT.log 2>&1 || exit 1
  python3 - "$OUT" "$l" <<'PY'
import csv, glob, sys
rows = []
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "walk_bounce" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
first = min(d for d, _, _ in rows)
second = sorted({d for d, _, _ in rows})[1]
for d in (first, second):
    print(sys.argv[2], "dispatch", d, {n: f"{v:.4g}" for dd, n, v in rows if dd == d})
PY
done
