#!/usr/bin/env python3
"""Wall-clock timing of the whole hot path through the C ABI (index build, phases 1-4) on one workload.
usage: python tools/pipeline_time.py [Cm|small|tiny|example:<name>] [seed] [reps] [base attempts per trial, default 100]"""
import json
import os
import sys
import time

# The host-side bookkeeping around the trials is a few tiny numpy calls: keep the BLAS / OpenMP pools to one thread.  On the GPU boxes
# the process sees 256 CPUs but its cgroup has a quota of 16; OpenBLAS sizes its pool by the former, its spinning workers exhaust the
# latter, and the kernel then freezes the whole process for the rest of a 100 ms period -- the sporadic 65-80 ms "stall" of rounds
# 1-3 (profiles/r03_stall_root_cause.json).  Must happen before numpy is imported.
for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import synth  # noqa: E402
from model_matching_amd.estimator import StocsEstimator  # noqa: E402


def main():
    if os.environ.get("PT_TORCH"):      # A/B: the same trials in a process that has torch's HIP runtime state (bench.py has)
        import torch
        torch.cuda.init()
        _keep = torch.zeros(1 << 20, device="cuda")
        if os.environ["PT_TORCH"] == "2":
            torch.cuda.set_stream(torch.cuda.Stream())
    name = sys.argv[1] if len(sys.argv) > 1 else "Cm"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    attempts = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    if name.startswith("example:"):
        d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "example_%s.npz" % name.split(":")[1]))
        args = (d["scene_pos"], d["scene_nrm"], d["scene_prob"], d["scene_pixel"], d["model_pos"], d["model_nrm"])
        T_gt = None
    else:
        m, s, k = synth.workload(name)
        args = (s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm)
        T_gt = s.T_gt
    t = time.perf_counter()
    est = StocsEstimator(*args, build_index=True)
    est.sync()
    t_ctx = time.perf_counter() - t
    rep = {"workload": name, "nS": est.nS, "nM": est.nM, "ctx_create_incl_index_s": t_ctx, "index": est.index_stats(), "runs": []}
    if os.environ.get("PT_BURN") and T_gt is not None:      # A/B: what bench.py has done to the chip before its pipeline section -- PT_BURN launches of the 65 536-candidate metric batch on a second context
        est2 = StocsEstimator(*args, build_index=False)
        T = synth.make_candidates(T_gt, k)
        for _ in range(int(os.environ["PT_BURN"])):
            est2.score_transforms(T)
        if os.environ.get("PT_BURN_CLOSE"):
            est2.close()
    for r in range(reps):
        est.L.stocs_clear_bases(est.h)
        t0 = time.perf_counter()
        valid, ids, inv = est.sample_bases(seed + r, attempts)
        t1 = time.perf_counter()
        nq = est.find_congruent_all()
        t2 = time.perf_counter()
        nc = est.make_transforms(200, seed + r)
        t3 = time.perf_counter()
        best_lcp, best_idx, pose = est.compute_best_transform()
        t4 = time.perf_counter()
        run = {"bases": int(valid.sum()), "quads": int(nq), "candidates": int(nc), "best_lcp": float(best_lcp),
               "t_sample_ms": (t1 - t0) * 1e3, "t_congruent_ms": (t2 - t1) * 1e3, "t_transforms_ms": (t3 - t2) * 1e3,
               "t_verify_ms": (t4 - t3) * 1e3, "poses_per_s_phases_2_4": nc / max(t4 - t1, 1e-9)}
        if T_gt is not None and best_idx >= 0:
            P = pose.reshape(4, 4).T
            dR = P[:3, :3].T @ T_gt[:3, :3]
            run["rot_err_deg"] = float(np.degrees(np.arccos(min(1.0, (np.trace(dR) - 1) / 2))))
            run["tr_err_mm"] = float(np.linalg.norm(P[:3, 3] - T_gt[:3, 3]) * 1e3)
        run["steps_ms"] = {c: est.last_call_timing(w) for w, c in enumerate(("find_congruent_all", "make_transforms", "verify_all"))}
        rep["runs"].append(run)
    if len(rep["runs"]) > 2:   # the library's own step record, median over the runs after the first two
        rep["steps_ms_median"] = {c: [[lab, float(np.median([dict(r["steps_ms"][c]).get(lab, 0.0) for r in rep["runs"][2:]]))] for lab, _ in rep["runs"][-1]["steps_ms"][c]]
                                  for c in rep["runs"][-1]["steps_ms"]}
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
