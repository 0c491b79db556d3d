#!/usr/bin/env python3
"""Counters of EVERY kernel of one command: a --kernel-trace --stats pass for the durations, then the --pmc passes of
tools/pmc.py (one per counter group, never combined with other traces), derived per kernel name.

    python tools/pmc_all.py <outdir> -- python3 tools/pipeline_time.py Cm 1234 4   ->  JSON, one record per kernel"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc  # noqa: E402


def short(name):
    n = name.split("(")[0]
    for pre in ("void ", "stocs::", "rocprim::ROCPRIM_400200_NS::detail::"):
        n = n.replace(pre, "")
    return n[:80]


def stats_pass(cmd, outdir, timeout=300):
    os.makedirs(outdir, exist_ok=True)
    full = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", outdir, "--"] + list(cmd)
    with open(os.path.join(outdir, "log"), "w") as log:
        p = subprocess.Popen(full, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
        try:
            p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, 9)
            p.wait()
            return {}
    out = {}
    for f in glob.glob(os.path.join(outdir, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Name"]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])}
    return out


def main():
    i = sys.argv.index("--")
    outdir, cmd = os.path.abspath(sys.argv[1]), sys.argv[i + 1:]   # rocprofv3 runs from /tmp: relative paths would land there
    t0 = time.time()
    stats = stats_pass(cmd, os.path.join(outdir, "stats"))
    per = collections.defaultdict(lambda: {"counters": {}, "res": {}})
    passes = {}
    for g, counters in pmc.GROUPS.items():
        f, status = pmc.run_pass(counters, cmd, os.path.join(outdir, g), 300)
        passes[g] = status
        if not f:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[k]["res"] = {"vgpr": r.get("VGPR_Count"), "accum_vgpr": r.get("Accum_VGPR_Count"), "sgpr": r.get("SGPR_Count"), "lds_block": r.get("LDS_Block_Size"),
                             "scratch": r.get("Scratch_Size"), "workgroup": r.get("Workgroup_Size"), "grid": r.get("Grid_Size")}
        for k, cs in acc.items():
            for name, v in cs.items():
                rec = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
                if name == "GRBM_GUI_ACTIVE":
                    per[k]["counters"]["GRBM_GUI_ACTIVE@" + g] = rec
                    per[k]["counters"].setdefault(name, rec)
                else:
                    per[k]["counters"][name] = rec
    out = []
    for k, v in per.items():
        st = stats.get(k) or stats.get(k.split("(")[0]) or {}
        ms = st.get("avg_us", 0) / 1e3 or None
        d = pmc.derive(v, ms)
        c = {n: x["per_launch_mean"] for n, x in v["counters"].items()}
        out.append({"kernel": short(k), "calls": st.get("calls"), "avg_us": st.get("avg_us"), "total_ms": st.get("total_ms"), "pct_of_gpu_time": st.get("pct"),
                    "resources": v["res"], "derived": {a: b for a, b in d.items() if a != "binding"}, "binding": d.get("binding"),
                    "valu_insts": c.get("SQ_INSTS_VALU"), "vmem_rd_insts": c.get("SQ_INSTS_VMEM_RD"), "lds_insts": c.get("SQ_INSTS_LDS")})
    out.sort(key=lambda r: -(r["total_ms"] or 0))
    print(json.dumps({"command": " ".join(cmd), "passes": passes, "seconds": time.time() - t0, "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
