#!/usr/bin/env python3
"""Kernel time of one scoring launch (HIP events, stocs_time_score_kernel) with a chosen build of the library -- for A/B runs of
two builds on one box, one process each.  usage: python tools/lcp_time.py <lib.so> [Cm|C5] [rounds] [candidates] [key=value options]"""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import capi, synth  # noqa: E402

capi.LIB_PATH = os.path.abspath(sys.argv[1])
_probe = ctypes.CDLL(capi.LIB_PATH)
for name in list(capi.SIGNATURES):   # an older build lacks the newest entry points
    if not hasattr(_probe, name):
        del capi.SIGNATURES[name]
from model_matching_amd.estimator import StocsEstimator  # noqa: E402

name = sys.argv[2] if len(sys.argv) > 2 else "Cm"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
m, s, k = synth.workload(name)
if len(sys.argv) > 4 and int(sys.argv[4]) > 0:
    k = int(sys.argv[4])
est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
for kv in sys.argv[5:]:
    key, v = kv.split("=")
    est.set_option(key, int(v))
cs, cm = est.get_scene_centroid().astype(np.float64), est.get_model_centroid().astype(np.float64)
T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), k)
dT, dL = est.dev_alloc(T.nbytes), est.dev_alloc(k * 4)
est.dev_upload(dT, T)
est.score_device(dT, k, dL)
out = np.zeros(k, np.float32)
est.dev_download(dL, out)
t = [est.time_score_kernel(dT, k, dL, 10) for _ in range(rounds)]
print(json.dumps({"lib": os.path.basename(capi.LIB_PATH), "workload": name, "K": k, "options": sys.argv[5:], "ms_median": float(np.median(t)), "ms_all": t,
                  "Mposes_per_s": k / float(np.median(t)) / 1e3, "score_sum": float(out.astype(np.float64).sum()), "score_crc": int(np.bitwise_xor.reduce(out.view(np.uint32)))}))
