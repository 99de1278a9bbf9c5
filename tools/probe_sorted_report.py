"""python3 tools/probe_sorted_report.py DIR LOG: search_kernel durations from a rocprofv3 kernel trace, paired with the labels in LOG."""
import csv, glob, sys
rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "search_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
ms = [d for _, d in rows]
labels = [l.split(" ", 1)[1].strip() for l in open(sys.argv[2]) if l.startswith("order ")]
print("launches", len(ms))
# launches: 2 segments of set-up use the thread route (no search_kernel); then 1 (base) + 2 per order
ms = ms[1:]
for i, name in enumerate(labels):
    pair = ms[2 * i: 2 * i + 2]
    print(f"{name:18s} search_kernel ms {min(pair):.3f}")
