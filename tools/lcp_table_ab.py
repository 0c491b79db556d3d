#!/usr/bin/env python3
"""Interleaved A/B of the LCP kernel over the forms of the flat cell table: one context per form (the table is built with the
scene grid), the same resident candidate batch, HIP events on each context's stream, several rounds; scores must be bitwise
equal.  usage: python tools/lcp_table_ab.py [Cm|small] [rounds] [candidates] [forms, e.g. 1,2,0]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_matching_amd import synth  # noqa: E402
from model_matching_amd.estimator import StocsEstimator  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "Cm"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
m, s, k = synth.workload(name)
if len(sys.argv) > 3 and int(sys.argv[3]) > 0:
    k = int(sys.argv[3])
forms = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,2,0").split(",")]
labels = {0: "brick look-ups (top -> cell word)", 1: "flat table, row-major", 2: "flat table, 2x2x2 blocks"}
ests, bufs, res, times = {}, {}, {}, {}
T = None
for f in forms:
    est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
    est.set_option("lcp_flat", f)
    est.set_scene(s.pos, s.nrm, s.prob, s.pixel)    # rebuilds the grid with the chosen table
    if T is None:
        cs, cm = est.get_scene_centroid().astype(np.float64), est.get_model_centroid().astype(np.float64)
        T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), k)
    dT, dL = est.dev_alloc(T.nbytes), est.dev_alloc(k * 4)
    est.dev_upload(dT, T)
    est.score_device(dT, k, dL)
    out = np.zeros(k, np.float32)
    est.dev_download(dL, out)
    ests[f], bufs[f], res[f], times[f] = est, (dT, dL), out, []
for r in range(rounds):
    for f in forms:
        times[f].append(ests[f].time_score_kernel(bufs[f][0], k, bufs[f][1], 20))
print(json.dumps({"workload": name, "K": k, "forms": {labels[f]: {"median_ms": float(np.median(times[f])), "all_ms": times[f]} for f in forms},
                  "bitwise_equal": bool(all(np.array_equal(res[forms[0]], res[f]) for f in forms))}))
