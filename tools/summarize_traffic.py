"""python3 tools/summarize_traffic.py DIR ROUND: per-kernel FETCH_SIZE / WRITE_SIZE sums of the two
rocprofv3 --pmc passes under DIR -> profiles/rR_pmc_{fetch,write}.csv and profiles/hbm_traffic.json."""
import collections
import csv
import glob
import json
import sys

d, rnd = sys.argv[1], sys.argv[2]
KERNEL = "search_kernel"


def per_kernel(sub, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            agg[name][0] += 1
            agg[name][1] += float(r["Counter_Value"])
    return agg


out = {}
for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    agg = per_kernel(sub, counter)
    with open(f"profiles/r{rnd}_pmc_{sub}.csv", "w") as f:
        f.write(f"kernel,launches,{counter}_KiB_sum,{counter}_KiB_per_launch\n")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k},{n},{v:.1f},{v / n:.1f}\n")
    hit = [(n, v) for k, (n, v) in agg.items() if KERNEL in k]
    n, v = hit[0]
    out[counter] = {"launches": n, "sum_KiB": v, "mean_KiB_per_launch": v / n}
fetch = out["FETCH_SIZE"]["mean_KiB_per_launch"] * 1024
write = out["WRITE_SIZE"]["mean_KiB_per_launch"] * 1024
json.dump({
    "round": int(rnd),
    "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py "
               "--steps 1 --warmup 0 --no-cpu-baseline",
    "kernel": "ptk::" + KERNEL,
    "workload": "dragon stand-in 1920x1080 64spp depth 8",
    "counters": out,
    "fetch_bytes_per_launch_raw": fetch,
    "write_bytes_per_launch": write,
    "bytes_per_launch": 2 * fetch + write,
    "note": "FETCH_SIZE is doubled as prescribed for gfx950 (128-B requests tallied at 64 B); the factor is calibrated "
            "for wide coalesced reads, the 64-B node and 48-B triangle records read here are uncalibrated, so the true "
            "figure lies between raw+write and 2*raw+write.",
}, open("profiles/hbm_traffic.json", "w"), indent=1)
print(json.dumps(out))
