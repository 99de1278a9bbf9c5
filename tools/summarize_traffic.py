"""python3 tools/summarize_traffic.py DIR ROUND [flattened]: per-kernel FETCH_SIZE / WRITE_SIZE sums of the two
rocprofv3 --pmc passes under DIR/fetch and DIR/write -> profiles/rR_pmc_{fetch,write}[_flattened].csv and
profiles/hbm_traffic[_flattened].json (bytes per search_kernel launch, which bench.py reports as `traffic`)."""
import collections
import csv
import glob
import json
import sys

d, rnd = sys.argv[1], sys.argv[2]
flattened = len(sys.argv) > 3 and sys.argv[3] == "flattened"
suffix = "_flattened" if flattened else ""
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
    with open(f"profiles/r{rnd}_pmc_{sub}{suffix}.csv", "w") as f:
        f.write(f"kernel,launches,{counter}_KiB_sum,{counter}_KiB_per_launch\n")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k},{n},{v:.1f},{v / n:.1f}\n")
    hit = [(n, v) for k, (n, v) in agg.items() if KERNEL in k]
    n = sum(h[0] for h in hit)
    v = sum(h[1] for h in hit)
    out[counter] = {"launches": n, "sum_KiB": v, "mean_KiB_per_launch": v / n}
renders = 2 if flattened else 1  # tools/probe_flattened.py renders the frame twice, tools/profile_render.py once
fetch = out["FETCH_SIZE"]["mean_KiB_per_launch"] * 1024
write = out["WRITE_SIZE"]["mean_KiB_per_launch"] * 1024
if flattened:
    command = "python3 -u tools/probe_flattened.py 24 3840 2160 2 0"
    workload = "24 flattened copies (2.1e7 world-space triangles in one mesh) 3840x2160 2spp depth 8"
    # every search fetches its own records from HBM here (nothing is re-used from cache), and the raw counter equals the
    # bytes of the 64-B sectors those records occupy: the gfx950 doubling for wide coalesced reads does not apply
    total = fetch + write
    note = ("FETCH_SIZE taken raw: the 64-B node and 48-B triangle records are fetched as single 64-B sectors, for which "
            "the counter's 64 B per request is the true size (the x2 correction is for 128-B requests of wide coalesced reads)")
else:
    command = "python3 tools/profile_render.py 64"
    workload = "dragon stand-in 1920x1080 64spp depth 8"
    total = 2 * fetch + write
    note = ("FETCH_SIZE is doubled as prescribed for gfx950 (128-B requests tallied at 64 B); the factor is calibrated "
            "for wide coalesced reads, the 64-B node and 48-B triangle records read here are uncalibrated, so the true "
            "figure lies between raw+write and 2*raw+write.  The counter includes Infinity-Cache hits.")
try:
    fingerprint = open(f"{d}/kernel_fingerprint.txt").read().strip()
except OSError:
    fingerprint = None
json.dump({
    "round": int(rnd),
    "kernel_fingerprint": fingerprint,  # device code the passes ran (pathtracing_amd._native.kernel_fingerprint)
    "command": f"rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- {command}",
    "kernel": "ptk::" + KERNEL,
    "workload": workload,
    "counters": out,
    "fetch_bytes_per_launch_raw": fetch,
    "write_bytes_per_launch": write,
    "bytes_per_launch": total,
    # per rendered frame (the launches of one frame together): what bench.py divides by its own launch count
    "bytes_per_frame": total * out["FETCH_SIZE"]["launches"] / renders,
    "note": note,
}, open(f"profiles/hbm_traffic{suffix}.json", "w"), indent=1)
print(json.dumps(out))
