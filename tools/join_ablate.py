#!/usr/bin/env python3
"""Prices the two halves of join_count_kernel (cone sampling / walk over the P run) by switching them off (measurement build,
STOCS_JOIN_ABLATE; counts of ablated runs are wrong by design).  Device time of the join group from the library's own step record."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import capi, synth  # noqa: E402
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libstocs_hip_tools.so")
from model_matching_amd.estimator import StocsEstimator  # noqa: E402

m, s, k = synth.workload("Cm")
est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=True)
est.set_option("device_clock", 1)   # the "device: ..." steps (HIP events between the kernel groups) are opt-in since round 5b
out = {}
for abl, what in ((0, "full"), (1, "no cone sampling"), (2, "no walk over the P run"), (3, "neither")):
    os.environ["STOCS_JOIN_ABLATE"] = str(abl)
    ts = []
    for r in range(10):
        est.L.stocs_clear_bases(est.h)
        est.sample_bases(1234 + r % 5, 100)
        est.find_congruent_all()
        ts.append(dict(est.last_call_timing(0))["device: join count"])
    out[what] = float(np.median(ts[2:]))
os.environ["STOCS_JOIN_ABLATE"] = "0"
sweep = {}
for gmin, rmin in ((1, 64), (1, 32), (1, 48), (1, 96), (1, 128), (1, 16), (2, 64), (4, 64), (12, 48), (65, 0)):
    os.environ.update(STOCS_JOIN_GMIN=str(gmin), STOCS_JOIN_RMIN=str(rmin))
    ts = []
    for r in range(10):
        est.L.stocs_clear_bases(est.h)
        est.sample_bases(1234 + r % 5, 100)
        est.find_congruent_all()
        ts.append(dict(est.last_call_timing(0))["device: join count"])
    sweep["group>=%d run>=%d" % (gmin, rmin)] = float(np.median(ts[2:]))
print(json.dumps({"device_ms_of_the_join_count_group_median_of_8_trials": out, "threshold_sweep": sweep}, indent=1))
