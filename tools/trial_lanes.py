#!/usr/bin/env python3
"""N trials of a frame as L concurrent batches: L contexts, L host threads, each one stocs_run_trials call over its share of the seeds.
What the phases of different batches gain from running side by side (the join of one next to the gathers of another): Cm 64 trials
1 690 -> 1 810 trials/s with two lanes, ycb 20 900 -> 24 200, packed dove 5 520 -> 5 780; three and four lanes give less.
usage: python tools/trial_lanes.py [Cm|ycb_024_bowl|linemod_obj_06|packed_dove] [trials] [lanes]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("STOCS_PIN_BLAS", "1")
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from model_matching_amd import synth
from model_matching_amd.estimator import StocsEstimator
name = sys.argv[1] if len(sys.argv) > 1 else "Cm"
n_trials = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if name == "Cm":
    m, s, k = synth.workload("Cm"); args = (s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm); mode = 0; edge = None
else:
    d = np.load(os.path.join(ROOT, "tests", "golden", "example_%s.npz" % name))
    args = (d["scene_pos"], d["scene_nrm"], d["scene_prob"], d["scene_pixel"], d["model_pos"], d["model_nrm"]); mode = 1 if "edge_map" in d.files else 0
    edge = d["edge_map"] if mode else None
ests = [StocsEstimator(*args, build_index=True) for _ in range(lanes)]
if mode:
    for e in ests: e.set_edge_map(edge)
seeds = list(range(3, 3 + n_trials))
parts = [seeds[i::lanes] for i in range(lanes)]
def run(i):
    return ests[i].run_trials(parts[i], 100, mode=mode, dispersion=0.9, max_per_base=200)
with ThreadPoolExecutor(lanes) as ex:
    list(ex.map(run, range(lanes)))     # sizes the arenas
    for e in ests: e.sync()
    best = 0
    for rep in range(3):
        t0 = time.perf_counter()
        res = list(ex.map(run, range(lanes)))
        dt = time.perf_counter() - t0
        best = max(best, n_trials / dt)
print(json.dumps({"example": name, "trials": n_trials, "lanes": lanes, "trials_per_s_best_of_3": best}))
