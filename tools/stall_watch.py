#!/usr/bin/env python3
"""Runs N Cm trials on one warm context and prints, for every trial whose stocs_find_congruent_all / stocs_make_transforms /
stocs_verify_all took more than 5x the median, the library's own step record of that call (stocs_last_call_timing) -- the
80 ms stalls of rounds 1-2 were the same trial (seed 1234 + 3) both times.  usage: python tools/stall_watch.py [trials] [first seed]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import synth  # noqa: E402
from model_matching_amd.estimator import StocsEstimator  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
period = int(sys.argv[3]) if len(sys.argv) > 3 else 8      # trials cycle through this many seeds (0: every trial its own seed)
m, s, k = synth.workload("Cm")
# what bench.py has in its process before its pipeline section, switchable, to find what the sporadic ~70 ms pause follows:
# SW_TORCH=1 PyTorch's HIP runtime initialised first; SW_SCORE=1 a first context that has scored batches on torch's stream
if os.environ.get("SW_TORCH") == "1":
    import torch
    torch.cuda.set_device(0)
    _x = torch.zeros(1, device="cuda")
if os.environ.get("SW_SCORE") == "1":
    est0 = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
    if os.environ.get("SW_TORCH") == "1":
        est0.set_stream(torch.cuda.current_stream().cuda_stream)
    cs0, cm0 = est0.get_scene_centroid().astype(np.float64), est0.get_model_centroid().astype(np.float64)
    T0 = synth.make_candidates(synth.centred_gt(s.T_gt, cs0, cm0), k)
    dT0, dL0 = est0.dev_alloc(T0.nbytes), est0.dev_alloc(k * 4)
    est0.dev_upload(dT0, T0)
    for _ in range(int(os.environ.get("SW_SCORE_STEPS", "300"))):
        est0.score_device(dT0, k, dL0)
    est0.sync() if os.environ.get("SW_TORCH") != "1" else torch.cuda.synchronize()
est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=True)
rows = []
# SW_CPUWORK=1: what bench.py does on the host between two trials of its pipeline section.  Together with STOCS_KEEP_BLAS_THREADS=1
# (BLAS pool left at its default size) this reproduces the 65-80 ms pause: profiles/r03_stall_root_cause.json
cpu_work = os.environ.get("SW_CPUWORK") == "1"
if cpu_work:
    from scipy.spatial import cKDTree
    gt_pts = m.pos.astype(np.float64) @ np.asarray(s.T_gt, np.float64)[:3, :3].T + np.asarray(s.T_gt, np.float64)[:3, 3]
    gt_tree = cKDTree(gt_pts)
for r in range(n):
    est.L.stocs_clear_bases(est.h)
    a0 = int(est.L.stocs_device_alloc_count())
    t0 = time.perf_counter(); valid, _, _ = est.sample_bases(seed0 + (r % period if period else r), 100)
    t1 = time.perf_counter(); nq = est.find_congruent_all()
    t2 = time.perf_counter(); nc = est.make_transforms(200, seed0 + (r % period if period else r))
    t3 = time.perf_counter(); est.compute_best_transform()
    t4 = time.perf_counter()
    if cpu_work:
        P = est.best_pose.reshape(4, 4).T.astype(np.float64)
        est_pts = m.pos.astype(np.float64) @ P[:3, :3].T + P[:3, 3]
        _ = float(gt_tree.query(est_pts)[0].mean())
    rows.append({"trial": r, "seed": seed0 + (r % period if period else r), "bases": int(valid.sum()), "quads": int(nq), "candidates": int(nc),
                 "ms": [(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3], "allocs": int(est.L.stocs_device_alloc_count()) - a0,
                 "steps": [est.last_call_timing(w) for w in range(3)]})
warm = rows[(period or 2):]
med = [float(np.median([x["ms"][i] for x in warm])) for i in range(4)]
flag = [x for x in warm if any(x["ms"][i] > 5 * med[i] for i in range(1, 4))]
print(json.dumps({"trials": n, "median_ms_sample_congruent_transforms_verify": med, "max_ms": [float(max(x["ms"][i] for x in warm)) for i in range(4)],
                  "allocations_in_warm_trials": int(sum(x["allocs"] for x in warm)), "stalled_trials": flag,
                  "example_steps_of_a_normal_trial": warm[0]["steps"],
                  "all_trials_seed_quads_congruent_ms": [[x["seed"], x["quads"], round(x["ms"][1], 3)] for x in rows]}, indent=1))
