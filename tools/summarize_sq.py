"""python3 tools/summarize_sq.py DIR ROUND: SQ counters of the trace kernels (one rocprofv3 --pmc pass under DIR) ->
profiles/rR_sq_counters.txt (sums over launches) and profiles/sq_counters.json (per launch: wave64 vector instructions
issued, share of lanes active in them), which bench.py turns into the issue-rate record beside its roofline."""
import collections
import csv
import glob
import json
import sys

d, rnd = sys.argv[1], sys.argv[2]
KERNELS = ("search_kernel", "walk_bounce_kernel", "camera_kernel", "accumulate_kernel")
agg = {k: collections.defaultdict(float) for k in KERNELS}
launches = {k: collections.defaultdict(int) for k in KERNELS}
for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]] += 1
out = {}
with open(f"profiles/r{rnd}_sq_counters.txt", "w") as f:
    f.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES "
            "SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -- python3 tools/profile_render.py 64\n")
    f.write("# dragon stand-in 1920x1080, 64 spp, depth 8, one render; sums over launches\n")
    for k in KERNELS:
        if not agg[k]:
            continue
        f.write(f"[{k}]\n")
        for c in sorted(agg[k]):
            f.write(f"{c:32s} launches {launches[k][c]:4d} sum {agg[k][c]:.6g}\n")
        n = launches[k].get("SQ_INSTS_VALU", 0)
        if n:
            valu = agg[k]["SQ_INSTS_VALU"]
            out[k] = {"launches": n, "valu_insts_per_launch": valu / n, "valu_insts_per_frame": valu,
                      "active_lane_share": agg[k]["SQ_THREAD_CYCLES_VALU"] / (64.0 * valu) if valu else None,
                      "salu_insts_per_launch": agg[k]["SQ_INSTS_SALU"] / n, "vmem_rd_insts_per_launch": agg[k]["SQ_INSTS_VMEM_RD"] / n,
                      "wait_any_share_of_wave_cycles": agg[k]["SQ_WAIT_ANY"] / agg[k]["SQ_WAVE_CYCLES"] if agg[k]["SQ_WAVE_CYCLES"] else None}
import os  # noqa: E402

up = os.path.dirname(d.rstrip("/"))
try:
    out["kernel_fingerprint"] = open(f"{up}/kernel_fingerprint.txt").read().strip()  # device code the pass ran
except OSError:
    out["kernel_fingerprint"] = None
try:
    # static share of each kernel's vector instructions in the slow issue class (tools/issue_mix.py, same library)
    for name, rec in json.load(open(f"{up}/issue_mix.json")).items():
        for k in KERNELS:
            if k in name and k in out:
                out[k]["slow_class_share"] = rec["slow_class_share"]
                out[k]["mean_issue_cycles"] = rec["mean_issue_cycles"]
except (OSError, ValueError):
    pass
out["round"] = int(rnd)
out["workload"] = "dragon stand-in 1920x1080 64spp depth 8, one render"
json.dump(out, open("profiles/sq_counters.json", "w"), indent=1)
print(open(f"profiles/r{rnd}_sq_counters.txt").read())
