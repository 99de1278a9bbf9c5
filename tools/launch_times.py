"""Per-launch durations from a rocprofv3 kernel trace: python3 tools/launch_times.py DIR"""
import csv
import glob
import sys

rows = []
for path in glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0 = rows[0][0] if rows else 0
for s, e, name in rows:
    short = name.split("(")[0].replace("void ", "").replace("ptk::", "")[:28]
    print(f"{(s - t0) / 1e6:10.3f} ms  {short:28s} {(e - s) / 1e6:8.3f} ms")
