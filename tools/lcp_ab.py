#!/usr/bin/env python3
"""Interleaved A/B timing of the LCP kernel variants in ONE process (HIP events on the context's
stream), plus a bitwise comparison of their scores.  usage: python tools/lcp_ab.py [Cm|C5|small] [rounds]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("STOCS_PIN_BLAS", "1")   # harness side: one BLAS thread under the cgroup CPU quota; the library import itself has no side effects
from model_matching_amd import capi, synth  # noqa: E402
# variants beyond 0 / 15 / 24 / 31 exist only in the measurement build (make -C model_matching_amd/csrc tools)
_tools_lib = os.path.join(os.path.dirname(capi.LIB_PATH), "libstocs_hip_tools.so")
if os.path.exists(_tools_lib):
    capi.LIB_PATH = _tools_lib
from model_matching_amd.estimator import StocsEstimator  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "Cm"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    variants = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1"])]
    m, s, k = synth.workload(name)
    est = StocsEstimator(s.pos, s.nrm, s.prob, s.pixel, m.pos, m.nrm, build_index=False)
    cs = est.get_scene_centroid().astype(np.float64)
    cm = est.get_model_centroid().astype(np.float64)
    T = synth.make_candidates(synth.centred_gt(s.T_gt, cs, cm), k)
    dT, dL = est.dev_alloc(T.nbytes), est.dev_alloc(k * 4)
    est.dev_upload(dT, T)
    res, times = {}, {v: [] for v in variants}
    for v in variants:
        est.set_option("lcp_variant", v)
        est.score_device(dT, k, dL)
        out = np.zeros(k, np.float32)
        est.dev_download(dL, out)
        res[v] = out
    for r in range(rounds):
        for v in variants:
            est.set_option("lcp_variant", v)
            times[v].append(est.time_score_kernel(dT, k, dL, 10))
    base = res[variants[0]]
    rep = {"workload": name, "K": k, "nS": est.nS, "nM": est.nM}
    for v in variants:
        t = np.array(times[v])
        rep["v%d" % v] = {"ms_median": float(np.median(t)), "ms_min": float(t.min()), "Mposes_per_s": k / np.median(t) / 1e3,
                          "bitwise_equal_to_v%d" % variants[0]: bool(np.array_equal(res[v], base)),
                          "max_abs_diff": float(np.abs(res[v] - base).max())}
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
