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
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    for sub, name in (("stats", "final_bench"), ("stats_C5", "C5_bench"), ("stats_trials", "trials64_ycb_batch"), ("pmc_pipe", "pipeline_Cm"), ("pmc_trials", "trials16_Cm_batch")):
        try:
            shutil.copy(newest(os.path.join(src, sub, "**", "*_kernel_stats.csv")), os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
        except ValueError:
            print("no kernel stats under", sub)
    for f in sorted(glob.glob(os.path.join(src, "*.json"))):
        b = os.path.basename(f)
        if b in ("lcp_pmc.json",) or b.startswith("."):
            continue
        name = {"bench.json": "final_bench", "bench_C5.json": "C5_bench", "pmc_pipeline_Cm.json": "pipeline_Cm_pmc_all_kernels", "pmc_trials_Cm16.json": "trials16_Cm_batch_pmc_all_kernels"}.get(b, b[:-5])
        shutil.copy(f, os.path.join(dst, "%s_%s.json" % (tag, name)))
    # counter passes of the scoring kernel: raw per-launch means + the derived bounds (kernel time: rocprofv3's own average of the stats pass)
    import csv
    p = os.path.join(src, "lcp_pmc.json")
    if os.path.exists(p):
        raw = json.load(open(p))
        ms = None
        for r in csv.DictReader(open(newest(os.path.join(src, "stats", "**", "*_kernel_stats.csv")))):
            if "lcp_coopq_kernel<false" in r["Name"]:
                ms = float(r["AverageNs"]) * 1e-6
                break
        raw["kernel_ms_rocprofv3_stats_pass"] = ms
        raw["derived"] = pmc.derive(raw, ms)
        raw["note"] = ("rocprofv3 --pmc, one pass per counter group (tools/pmc.py, tools/profile_round.sh); FETCH_SIZE/WRITE_SIZE in KB; gfx950 correction "
                       "(MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 1/2 of wide reads -> x2; WRITE_SIZE exact; the sum is the L2's memory-side traffic "
                       "(Infinity-Cache hits included), named hbm_* in this tool's keys")
        json.dump(raw, open(os.path.join(dst, "%s_final_lcp_pmc.json" % tag), "w"), indent=1)
    print("copied", sorted(x for x in os.listdir(dst) if x.startswith(tag)))


if __name__ == "__main__":
    main()
