#!/usr/bin/env python3
"""Kernel time of one scoring launch under the environment's grid switches (STOCS_GRID_PRUNE / STOCS_GRID_DIV / ...), options as key=value.
usage: python tools/prune_sweep.py <workload> [candidates] [key=value ...]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STOCS_PIN_BLAS", "1")
os.environ["STOCS_DEBUG_TIMING"] = "1"
from model_matching_amd import synth  # noqa: E402
from model_matching_amd.estimator import StocsEstimator  # noqa: E402
name = sys.argv[1]
m, s, k = synth.workload(name)
rest = sys.argv[2:]
if rest and rest[0].isdigit():
    k = int(rest[0]); rest = rest[1:]
est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
for kv in rest:
    key, v = kv.split("="); est.set_option(key, int(v))
cs, cm = est.get_scene_centroid().astype(np.float64), est.get_model_centroid().astype(np.float64)
T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), k)
dT, dL = est.dev_alloc(T.nbytes), est.dev_alloc(k * 4)
est.dev_upload(dT, T)
for _ in range(3):
    est.score_device(dT, k, dL)
sc = np.zeros(k, np.float32); est.dev_download(dL, sc)
t = [est.time_score_kernel(dT, k, dL, 10) for _ in range(5)]
print(json.dumps({"workload": name, "K": k, "env": {k_: v for k_, v in os.environ.items() if k_.startswith("STOCS_GRID") or k_.startswith("STOCS_LCP")}, "options": rest,
                  "ms_median": round(float(np.median(t)), 4), "score_crc": int(np.bitwise_xor.reduce(sc.view(np.uint32)))}), flush=True)
