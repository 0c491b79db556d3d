#!/usr/bin/env python3
"""Copies the summaries of tools/profile_round.sh from gpurun_out/<tag>/ into profiles/ (tracked).
usage: python tools/collect_profiles.py r02"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc  # noqa: E402


def newest(pattern):
    """gpurun_out/ accumulates one sub-directory per profiler run: take the most recent match."""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(newest(os.path.join(src, "stats", "**", "*_kernel_stats.csv")), os.path.join(dst, "%s_final_bench_kernel_stats.csv" % tag))
    shutil.copy(newest(os.path.join(src, "pmc_pipe", "**", "*_kernel_stats.csv")), os.path.join(dst, "%s_pipeline_Cm_kernel_stats.csv" % tag))
    for f, name in (("bench.json", "final_bench"), ("bench_C5.json", "C5_bench"), ("frame_latency.json", "frame_latency"), ("sweep.json", "sweep"), ("trials64_s1.json", "trials64_streams1"),
                    ("trials64_s8.json", "trials64_streams8"), ("pipeline_Cm.json", "pipeline_Cm")):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s.json" % (tag, name)))
    # counter passes: raw per-launch means + the derived bounds (kernel time: rocprofv3's own average of the stats pass)
    import csv
    for f, name, kern, stats in (("lcp_pmc.json", "final_lcp_pmc", "lcp_coop", "stats"), ("join_pmc.json", "join_count_pmc", "join_count_kernel", "pmc_pipe")):
        p = os.path.join(src, f)
        if not os.path.exists(p):
            continue
        raw = json.load(open(p))
        ms = None
        for r in csv.DictReader(open(newest(os.path.join(src, stats, "**", "*_kernel_stats.csv")))):
            if kern in r["Name"]:
                ms = float(r["AverageNs"]) * 1e-6
                break
        raw["kernel_ms_rocprofv3_stats_pass"] = ms
        raw["derived"] = pmc.derive(raw, ms)
        raw["note"] = ("rocprofv3 --pmc, one pass per counter group (tools/pmc.py, tools/profile_round.sh); FETCH_SIZE/WRITE_SIZE in KB; gfx950 correction "
                       "(MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 1/2 of wide reads -> x2; WRITE_SIZE exact")
        json.dump(raw, open(os.path.join(dst, "%s_%s.json" % (tag, name)), "w"), indent=1)
    print("copied", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
