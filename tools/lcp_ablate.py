#!/usr/bin/env python3
"""Prices the parts of the LCP kernel by switching them off one at a time (measurement build only: libstocs_hip_tools.so,
`make -C model_matching_amd/csrc tools`; the scores of an ablated run are wrong by design).  One context, the same resident
batch, HIP events, interleaved rounds.  usage: python tools/lcp_ablate.py [Cm] [rounds] [candidates]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import capi, synth  # noqa: E402
_tools_lib = os.path.join(os.path.dirname(capi.LIB_PATH), "libstocs_hip_tools.so")
if not os.path.exists(_tools_lib):
    raise SystemExit("build the measurement library first: make -C model_matching_amd/csrc tools")
capi.LIB_PATH = _tools_lib
from model_matching_amd.estimator import StocsEstimator  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "Cm"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m, s, k = synth.workload(name)
if len(sys.argv) > 3 and int(sys.argv[3]) > 0:
    k = int(sys.argv[3])
est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
cs, cm = est.get_scene_centroid().astype(np.float64), est.get_model_centroid().astype(np.float64)
T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), k)
dT, dL = est.dev_alloc(T.nbytes), est.dev_alloc(k * 4)
est.dev_upload(dT, T)
dense_cases = [(0, "full kernel"), (8, "no normal test"), (256, "no chunk bounds: every list scanned to its end"), (4, "only the first trip (two lines) of every list"),
               (2, "cell words read, nobody survives"), (128, "top table only (no cell words), nobody survives"), (1, "no look-ups at all")]
cases = [(0, "full kernel"), (32, "no scene-normal gather"), (16, "no model-normal gather"), (48, "no normal gathers"), (8, "no normal test at all"),
         (4, "no list loads"), (4 | 8, "no list loads, no normal test"), (2, "cell look-up done, nobody survives"), (1, "no cell look-up, nobody survives"),
         (64, "no patch test (every 64-point step walked)"), (1 | 64, "no patch test, no cell look-up")]
if name in ("C5", "dense"):   # the dense default is the queue kernel with early exit: its switches are those of the Cm list
    cases = [(0, "full kernel"), (48, "no normal gathers"), (8, "no normal test at all"), (4, "no list loads"), (4 | 8, "no list loads, no normal test"),
             (2, "cell look-up done, nobody survives"), (1, "no cell look-up, nobody survives")]
    if len(sys.argv) > 4 and sys.argv[4] == "v31":
        cases = dense_cases
        est_variant = 31
times = {c: [] for c, _ in cases}
if "est_variant" in globals():
    est.set_option("lcp_variant", est_variant)
for r in range(rounds):
    for c, _ in cases:
        os.environ["STOCS_LCP_ABLATE"] = str(c)
        times[c].append(est.time_score_kernel(dT, k, dL, 10))
print(json.dumps({"workload": name, "K": k, "cases": [{"ablate": c, "what": w, "ms_median": float(np.median(times[c])), "ms_all": times[c]} for c, w in cases]}, indent=1))
