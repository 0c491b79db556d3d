#!/usr/bin/env python3
"""Interleaved A/B of the LCP kernel with / without the flat cell table (stocs_set_option "lcp_flat"), one process,
HIP events on the context's stream; scores must be bitwise equal.  usage: python tools/lcp_flat_ab.py [Cm|small] [rounds] [candidates] [option=lcp_flat]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import synth  # noqa: E402
from model_matching_amd.estimator import StocsEstimator  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "Cm"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
m, s, k = synth.workload(name)
if len(sys.argv) > 3:
    k = int(sys.argv[3])
opt = sys.argv[4] if len(sys.argv) > 4 else "lcp_flat"
on_value = int(sys.argv[5]) if len(sys.argv) > 5 else 1
off_value = int(sys.argv[6]) if len(sys.argv) > 6 else 0
est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
cs, cm = est.get_scene_centroid().astype(np.float64), est.get_model_centroid().astype(np.float64)
T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), k)
dT, dL = est.dev_alloc(T.nbytes), est.dev_alloc(k * 4)
est.dev_upload(dT, T)
res, times = {}, {0: [], 1: []}
for f in (0, 1):
    est.set_option(opt, on_value if f else off_value)
    est.score_device(dT, k, dL)
    out = np.zeros(k, np.float32)
    est.dev_download(dL, out)
    res[f] = out
for r in range(rounds):
    for f in (0, 1):
        est.set_option(opt, on_value if f else off_value)
        times[f].append(est.time_score_kernel(dT, k, dL, 20))
print(json.dumps({"workload": name, "K": k, "option": opt, "off_ms": float(np.median(times[0])), "on_ms": float(np.median(times[1])),
                  "bitwise_equal": bool(np.array_equal(res[0], res[1])), "off_all": times[0], "on_all": times[1]}))
