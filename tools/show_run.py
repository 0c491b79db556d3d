#!/usr/bin/env python3
"""Prints the timing / counter summaries of a tools/run_gpu.sh run: python tools/show_run.py <tag> [kernel substrings...]"""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
want = sys.argv[2:] or ["join_count", "lcp_"]
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", tag)
f = os.path.join(d, "pipe_untimed.json")
if os.path.exists(f):
    for x in json.load(open(f))["runs"]:
        print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in x.items() if k in ("bases", "quads", "candidates", "t_sample_ms", "t_congruent_ms", "t_transforms_ms", "t_verify_ms", "poses_per_s_phases_2_4")})
fs = glob.glob(os.path.join(d, "pipe_stats", "*", "*_kernel_stats.csv"))
if fs:
    for r in list(csv.DictReader(open(fs[0])))[:14]:
        print("%-95s calls %5s avg_us %9.1f pct %5s" % (r["Name"][:95], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
for sub in ("pipe_sq", "pipe_busy", "pipe_tcp"):
    fs = glob.glob(os.path.join(d, sub, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        if any(w in k for w in want):
            print(sub, k)
            print("   " + "  ".join("%s %.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
