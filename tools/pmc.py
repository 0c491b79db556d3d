#!/usr/bin/env python3
"""rocprofv3 counter collection for one kernel of one command, as separate --pmc passes (never combined with
sys / hip / hsa traces), and the bounds derived from them.  Used live by bench.py (roofline.traffic / roofline.binding of
the run that prints the bench line) and by tools/profile_round.sh for the summaries kept under profiles/.

    python tools/pmc.py <kernel substring> <outdir> -- python3 bench.py --steps 3 ...
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import time

# one pass per group: <= 8 SQ, <= 4 TCC (FETCH_SIZE costs 3 of them, WRITE_SIZE 2), 2 GRBM slots (MI355X_MICROARCH.md, PMC slots)
GROUPS = collections.OrderedDict([
    ("fetch", ["FETCH_SIZE", "GRBM_GUI_ACTIVE"]),
    ("write", ["WRITE_SIZE", "GRBM_GUI_ACTIVE"]),
    ("sq", ["SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VMEM_RD",
            "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE"]),
    ("mem", ["TA_TA_BUSY_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_REQ_sum", "TCC_READ_sum", "TCC_HIT_sum", "TCC_MISS_sum",
             "GRBM_GUI_ACTIVE"]),
])

CU_NUM, SIMD_NUM, XCD_NUM = 256, 1024, 8
L2_PEAK_BPS = 34.5e12     # MI355X_MICROARCH.md, L2 (per XCD): ~34.5 TB/s aggregate
HBM_PEAK_BPS = 8.0e12


def run_pass(counters, cmd, outdir, timeout):
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    full = ["rocprofv3", "--pmc"] + list(counters) + ["--kernel-trace", "--output-format", "csv", "-d", outdir, "--"] + list(cmd)
    t = time.time()
    with open(os.path.join(outdir, "log"), "w") as log:
        p = subprocess.Popen(full, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
        try:
            rc = p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, 9)          # the exact process group started above
            p.wait()
            return None, "timeout after %.0f s" % timeout
    if rc != 0:
        return None, "rocprofv3 exit code %d" % rc
    fs = glob.glob(os.path.join(outdir, "**", "*_counter_collection.csv"), recursive=True)
    if not fs:
        return None, "no counter file"
    return max(fs, key=os.path.getmtime), "%.1f s" % (time.time() - t)


def collect(kernel_substr, cmd, outdir, groups=GROUPS, timeout=150, budget_s=420):
    """-> {"kernel": name, "counters": {name: {"per_launch_mean", "launches"}}, "passes": {group: status}}"""
    out = {"kernel": None, "counters": {}, "passes": {}}
    t0 = time.time()
    for g, counters in groups.items():
        if time.time() - t0 > budget_s:
            out["passes"][g] = "skipped (time budget)"
            continue
        f, status = run_pass(counters, cmd, os.path.join(outdir, g), timeout)
        out["passes"][g] = status
        if not f:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                out["kernel"] = r["Kernel_Name"].split("(")[0]
        for k, v in acc.items():
            rec = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
            if k == "GRBM_GUI_ACTIVE":     # busy cycles of THIS pass (profiled passes run at slightly different clocks)
                out["counters"]["GRBM_GUI_ACTIVE@" + g] = rec
                out["counters"].setdefault(k, rec)
            else:
                out["counters"][k] = rec
    return out


def derive(pmc, kernel_ms):
    """Bounds of one launch from the counters (per-launch means) and the un-profiled kernel time of the same process."""
    c = {k: v["per_launch_mean"] for k, v in pmc["counters"].items()}
    d = {}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B -> x2 (guide, HBM section)
        d["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    cyc = c.get("GRBM_GUI_ACTIVE@sq", c.get("GRBM_GUI_ACTIVE"))
    if cyc:
        d["busy_cycles_per_xcd"] = cyc / XCD_NUM            # measured, no assumed clock (the counter sums the 8 XCDs)
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE@sq" in c:
        # one wave64 VALU instruction issues over 2 cycles of its SIMD-32 (guide, per-instruction constants): peak = 1 / 2 cyc / SIMD
        d["valu_issue_frac"] = c["SQ_INSTS_VALU"] * 2.0 / (SIMD_NUM * c["GRBM_GUI_ACTIVE@sq"] / XCD_NUM)
        d["wave_wait_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
        d["waves_per_simd_avg"] = c["SQ_WAVE_CYCLES"] * 4.0 / (SIMD_NUM * c["GRBM_GUI_ACTIVE@sq"] / XCD_NUM) if c.get("SQ_WAVE_CYCLES") else None
    if "TA_TA_BUSY_sum" in c and "GRBM_GUI_ACTIVE@mem" in c:
        d["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / (CU_NUM * c["GRBM_GUI_ACTIVE@mem"] / XCD_NUM)
    if "TCC_REQ_sum" in c and kernel_ms:
        # every TCC request moves one 128-byte line or a 64-byte half of it; the 64-byte reading is the lower bound
        lo, hi = c["TCC_REQ_sum"] * 64.0, c["TCC_REQ_sum"] * 128.0
        d["l2_bytes_per_launch_64B_128B"] = [lo, hi]
        d["l2_bw_frac_64B_128B"] = [lo / (kernel_ms * 1e-3) / L2_PEAK_BPS, hi / (kernel_ms * 1e-3) / L2_PEAK_BPS]
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if d.get("hbm_bytes_per_launch") is not None and kernel_ms:
        # what the L2s asked of the fabric (Infinity-Cache hits included: FETCH_SIZE counts the L2's memory-side requests) over
        # the kernel time, against the HBM peak: a kernel whose lists do not stay in L2 is bound here, not in the CUs
        d["hbm_fabric_bps"] = d["hbm_bytes_per_launch"] / (kernel_ms * 1e-3)
        d["hbm_fabric_frac"] = d["hbm_fabric_bps"] / HBM_PEAK_BPS
    cand = {"valu_issue": d.get("valu_issue_frac"), "texture_addresser": d.get("ta_busy_frac"),
            "l2_bandwidth": (d.get("l2_bw_frac_64B_128B") or [None, None])[1], "hbm_fabric": d.get("hbm_fabric_frac")}
    cand = {k: v for k, v in cand.items() if v is not None}
    if cand:
        name = max(cand, key=cand.get)
        d["binding"] = {"bound": name, "frac": cand[name], "all": cand,
                        "note": "largest of the measured unit utilisations; the rest of the time the waves wait on dependent "
                                "cache look-ups (wave_wait_frac)" if cand[name] < 0.8 else "this unit is the ceiling"}
    return d


def main():
    if "--" not in sys.argv or len(sys.argv) < 5:
        raise SystemExit(__doc__)
    i = sys.argv.index("--")
    kernel, outdir = sys.argv[1], os.path.abspath(sys.argv[2])
    pmc = collect(kernel, sys.argv[i + 1:], outdir)
    print(json.dumps(pmc, indent=1))


if __name__ == "__main__":
    main()
